"""BGZF compression on the device (jk_bgzf_deflate): the GPU-side replacement of the reference's
FileBGZF / bgzip_file sinks (/root/reference/src/io.h:150-236, /root/reference/src/hts.h:140-180).
PyTorch only provides the device buffers."""
import ctypes as C

from . import _abi


def bgzf_bound(n):
    return int(_abi.lib().jk_bgzf_bound(int(n)))


def bgzf_deflate(data, device=0, return_ms=False):
    """Compress ``data`` (bytes-like, or a uint8 CUDA tensor that stays where it is) into a complete BGZF
    file image; returns a uint8 CUDA tensor (and the kernels' device milliseconds if ``return_ms``)."""
    import torch
    L = _abi.lib()
    dev = torch.device("cuda", device)
    if isinstance(data, torch.Tensor):
        src = data.contiguous().view(torch.uint8)
        if src.device != dev:
            src = src.to(dev)
    else:
        host = torch.frombuffer(bytearray(bytes(data)), dtype=torch.uint8) if len(data) else torch.empty(0, dtype=torch.uint8)
        src = host.to(dev)
    n = src.numel()
    cap = bgzf_bound(n)
    dst = torch.empty(cap, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    out_bytes, ms = C.c_uint64(), C.c_double()
    _abi.check(L.jk_bgzf_deflate(device, src.data_ptr() if n else None, n, dst.data_ptr(), cap, C.byref(out_bytes), C.byref(ms)))
    out = dst[:out_bytes.value]
    return (out, ms.value) if return_ms else out
