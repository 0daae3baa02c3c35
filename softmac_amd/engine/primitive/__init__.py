from .primitive_base import Primitive
from .mesh import Mesh
from .primitives import Primitives
