"""distCUDA2 backed by libmsgs_hip.so (msgs_dist2_knn3, ms-gs_amd/csrc/knn.hip).  GPU-only, no fallback."""
import ctypes as C

import torch

from diff_gaussian_rasterization import _backend as _B


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """[P,3] float CUDA tensor -> [P] float32: mean squared distance to the 3 nearest other points
    (used as torch.clamp_min(distCUDA2(xyz), 1e-7) at /root/reference/scene/gaussian_model.py:199)."""
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2: points must live on a HIP device ('cuda'); there is no CPU path")
    pts = points.detach().to(torch.float32).contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError(f"distCUDA2: expected [P,3], got {tuple(points.shape)}")
    P = int(pts.shape[0])
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        scratch = torch.empty(int(_B.lib.msgs_knn_scratch_bytes(P)), dtype=torch.uint8, device=pts.device)
        stream = C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)
        _B.check(_B.lib.msgs_dist2_knn3(C.c_void_p(pts.data_ptr()), P, C.c_void_p(out.data_ptr()),
                                        C.c_void_p(scratch.data_ptr()), scratch.numel(), stream), "msgs_dist2_knn3")
    return out
