"""MI355X-native DEFLATE / inflate engine with the zlib_ng / gzip_ng / gzip_ng_threaded face of
pycompression/python-zlib-ng.  The arithmetic runs in hand-written HIP kernels behind the C ABI of
libzng_amd.so (include/zng_amd.h); there is no CPU fallback."""
__version__ = "0.1.0"
