from .primitive_cloth import Primitive_Cloth  # noqa: F401
