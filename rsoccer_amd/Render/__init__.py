from rsoccer_amd.Render.raster import FieldRaster, VSS_VIEW, SSL_VIEW  # noqa: F401
