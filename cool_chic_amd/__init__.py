"""cool_chic_amd - MI355X-native decoder for Cool-chic 5.0 `.cool` bitstreams.

The hot path (integer ARM/IFCE entropy model + range decoder, latent-pyramid upsampling, synthesis,
integer planes) is hand-written HIP for gfx950 in csrc/, behind the C ABI of include/ccd.h; this
package is the thin host-side mirror of the reference's decode surface:

    reference (coolchic.*)                         here (cool_chic_amd.*)
    bitstream.decode.decode_video / decode_frame   bitstream.decode.decode_video / decode_frame
    bitstream.component.coolchic.encode_decode_coolchic(mode="decode")
                                                   bitstream.component.coolchic.encode_decode_coolchic
    bitstream.header.header.{Video,Frame,CoolChic}Header
                                                   bitstream.header.{Video,Frame,CoolChic}Header
    io.io.save_frame_data_to_file, io.framedata.FrameData
                                                   io.save_frame_data_to_file, io.FrameData
"""
from ._lib import CcdError, lib  # noqa: F401
from .batch import DecodeBatch  # noqa: F401



def pool_trim(device: int = 0) -> None:
    """Returns the device / pinned blocks that destroyed batches left in the library's per-device cache to the HIP runtime
    (include/ccd.h: ccd_pool_trim; caps: CCD_POOL_MAX_MB, CCD_PINNED_POOL_MAX_MB).  The cache is invisible to PyTorch's
    allocator: call this before a memory-hungry torch workload shares the GPU."""
    lib().ccd_pool_trim(int(device))


__version__ = "0.1.0"
