from .soft_rasterize import soft_rasterize, soft_rasterize_raw, SoftRasterizeFunction
