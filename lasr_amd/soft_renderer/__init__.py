"""MI355X-backed mirror of the reference `soft_renderer` package
(/root/reference/third_party/softras/soft_renderer/__init__.py): same public names."""
from . import functional
from .mesh import Mesh
from .renderer import SoftRenderer
from .transform import Projection, LookAt, Look, Transform
from .lighting import AmbientLighting, DirectionalLighting, Lighting
from .rasterizer import SoftRasterizer

__version__ = '1.0.0'
