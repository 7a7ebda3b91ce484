"""image-codecs_amd -- Python view of the MI355X-native JPEG path (ctypes over the C-ABI).

The product is the shared library ``lib/libimagecodecs_mi355x.so`` built from ``csrc/``:
  * ``stbi_*``  the reference's public surface (include/image_api.h; reference definitions in
                convert.c:188-266, image_api.c:74-145, codec/jpeg_write.c:368-388),
  * ``mij_*``   the GPU back end's C-ABI (include/mij.h; replaces the kernel seam
                codec/jpeg.c:83-85),
  * ``mjh_*``   the host entropy decoder (csrc/jpeg_entropy.h; codec/jpeg.c:88-558,1119-1756).
This module only binds those entry points so that tests and bench.py read like calls into the
reference: same names, same argument meaning, same error behaviour.  It never computes pixels
itself and it has no CPU fallback: without the library (or, for decode, without a gfx950 GPU)
calls fail loudly.

Because the directory name contains a hyphen the package is imported through the
``image_codecs_amd`` shim at the repository root (``import image_codecs_amd as ica``).
"""
from .binding import (  # noqa: F401
    LIB_PATH,
    MijError,
    ImageDesc,
    Batch,
    Context,
    Encoder,
    PinnedBuffer,
    host_transform,
    emit_jpeg,
    mij_write_jpg_to_memory,
    mij_write_jpg_batch,
    HostDecoder,
    lib,
    build_library,
    stbi_failure_reason,
    stbi_info_from_memory,
    stbi_load,
    stbi_load_from_memory,
    stbi_load_from_callbacks,
    stbi_load_from_file,
    stbi_info,
    stbi_info_from_file,
    stbi_info_from_callbacks,
    stbi_write_jpg,
    stbi_load_16_from_memory,
    stbi_set_flip_vertically_on_load,
    stbi_write_jpg_to_memory,
    detile_coefficients,
    host_decode_staged,
    compact_offsets,
    expand_compact_region,
    decode_jpegs_multi,
    gpu_available,
)
from .synth import synth_rgb, synth_jpeg, synth_rgb_edges  # noqa: F401
