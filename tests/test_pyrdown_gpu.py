"""Frame::ComputeImagePyramid (reference src/Frame.cpp:74-81: cv::pyrDown per level) on packed device pyramids:
the one-launch-per-pyramid kernel (band buffers in LDS) and the one-launch-per-level kernels against the CPU
oracle, byte for byte, on batches — several band heights, 3..5 levels, odd heights, and shapes that are not
eligible for the fused kernel (they must fall back, with the same bytes)."""
import ctypes as C

import numpy as np
import pytest

from dsdtm_amd import capi

pytestmark = pytest.mark.gpu


def _run(gpu_ctx, imgs, levels, env):
    import torch
    n, h, w = imgs.shape
    ws, hs, ss, offs, nbytes = capi.pyramid_layout(w, h, levels)
    pitch = (nbytes + 255) // 256 * 256
    host = np.zeros((n, pitch), np.uint8)
    host[:, :w * h] = imgs.reshape(n, -1)
    # poison the levels that are to be written
    host[:, offs[1]:nbytes] = 0xA5
    dev = torch.from_numpy(host).to("cuda:0")
    wa, ha, sa = (C.c_int * levels)(*ws), (C.c_int * levels)(*hs), (C.c_int * levels)(*ss)
    oa = (C.c_size_t * levels)(*offs)
    import contextlib
    with (capi.debug_options(**env) if env else contextlib.nullcontext()):
        st = torch.cuda.current_stream()
        gpu_ctx.check(gpu_ctx.lib.dsdtm_pyrdown_batch_device(gpu_ctx.handle, dev.data_ptr(), pitch, n, levels, wa, ha, sa, oa,
                                                            st.cuda_stream))
        torch.cuda.synchronize()
    out = dev.cpu().numpy()
    return [[out[i, offs[l]:offs[l] + ws[l] * hs[l]].reshape(hs[l], ws[l]) for l in range(levels)] for i in range(n)], \
        out[:, nbytes:]


@pytest.mark.parametrize("shape,levels,n", [((480, 640), 4, 5), ((480, 640), 5, 3), ((480, 640), 3, 2),
                                            ((250, 640), 4, 3), ((960, 1280), 4, 2), ((37, 64), 3, 4),
                                            ((241, 323), 4, 2), ((60, 80), 4, 3), ((120, 160), 3, 40)])
def test_batched_pyramids_match_the_oracle(gpu_ctx_each, oracle, shape, levels, n):
    gpu_ctx = gpu_ctx_each                                 # the release library's own choice of kernel; then every variant on the diagnostic one
    rng = np.random.default_rng(shape[0] * 7 + levels)
    imgs = rng.integers(0, 256, (n,) + shape, dtype=np.uint8)
    imgs[0, :, :] = 255                                    # saturation: (sum + 128) >> 8 must stay 255
    want = []
    for i in range(n):
        pyr = [imgs[i]]
        for _ in range(1, levels):
            pyr.append(oracle.pyrdown(pyr[-1]))
        want.append(pyr)
    ref_tail = None
    # band heights: automatic, the smallest, one that does not divide the coarsest level, the whole level
    F = {"pyr_fused": 2}                                  # the fused kernel wherever the shape allows it
    envs = ({}, F, dict(F, pyr_band=2), dict(F, pyr_band=7), dict(F, pyr_band=100000), {"pyr_fused": 0}) if gpu_ctx.diag else ({},)
    for env in envs:
        got, tail = _run(gpu_ctx, imgs, levels, env)
        for i in range(n):
            for l in range(levels):
                assert np.array_equal(got[i][l], want[i][l]), f"{env}: image {i} level {l}"
        # nothing is written behind the last level
        assert not tail.any(), f"{env}: bytes behind the pyramid were written"
