"""semabs_amd — MI355X-native relevancy -> 3D-UNet inference path of semantic-abstraction.

Python here is host plumbing (tensors, streams, weight loading, the reference's call surface);
all arithmetic on the path runs in `libsemabs_hip.so` (hand-written HIP for gfx950) behind the
C ABI declared in `include/semabs.h`.  See DESIGN.md.
"""
__all__ = ["weights"]
