"""stainx_amd -- MI355X-native (gfx950) backend for stainx-style stain normalisation."""
__version__ = "0.1.0"
