from .shape_maker import Shapes
