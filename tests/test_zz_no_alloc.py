"""sgmx_compute_census_u8_dev_noalloc (include/sgmx.h): the census path for callers that cannot allocate.
Through the C ABI, against sgmx_compute_census_u8_dev on the same inputs, bit for bit, on two calls in a row
(the second with a smaller input: the static scratch keeps the first call's bytes), then the limits."""
import ctypes as C

import numpy as np
import pytest

from conftest import synth_pair

pytestmark = pytest.mark.gpu


def _both(sv, torch, pairs, D, paths, max_batch):
    n = len(pairs)
    H, W = pairs[0][0].shape
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, num_paths=paths, max_batch=max_batch))
    st = torch.cuda.current_stream().cuda_stream
    want = torch.full((n, H, W), -1, dtype=torch.int16, device="cuda")
    got = torch.full((n, H, W), -1, dtype=torch.int16, device="cuda")
    # the C ABI entry points themselves (ctypes), not the Python conveniences
    rc = e.lib.sgmx_compute_census_u8_dev(e.h, n, L.data_ptr(), R.data_ptr(), W, H, want.data_ptr(), st)
    assert rc == 0, e.lib.sgmx_last_error(e.h)
    rc = e.lib.sgmx_compute_census_u8_dev_noalloc(e.h, n, L.data_ptr(), R.data_ptr(), W, H, got.data_ptr(), st)
    assert rc == 0, e.lib.sgmx_last_error(e.h)
    torch.cuda.synchronize()
    e.close()
    return got.cpu().numpy().view(np.uint16), want.cpu().numpy().view(np.uint16)


def test_noalloc_equals_dev_entry_point_two_calls_second_smaller(oracle):
    import torch
    import stereo_vision_amd as sv
    assert sv._lib.load().sgmx_noalloc_scratch_bytes() == 64 << 20
    # first call: 3 pairs of 288x200 at D=64, 8 paths, in launch groups of 2 (29.5 MB per pair: 2 fit the 64 MiB, and
    # max_batch is 2 as well)
    big = [synth_pair(700 + i, 200, 288, 50) for i in range(3)]
    got, want = _both(sv, torch, big, 64, 8, 2)
    assert np.array_equal(got, want)
    assert np.array_equal(got[1], oracle.sgm_census(big[1][0], big[1][1], 64, 5, 3, 60, 8))
    # second call, smaller input and another layout (D=16 -> one packed pair per lane, 4 paths): stale bytes of the
    # first call lie all over the scratch
    small = [synth_pair(800 + i, 61, 97, 12) for i in range(2)]
    got, want = _both(sv, torch, small, 16, 4, 4)
    assert np.array_equal(got, want)
    assert np.array_equal(got[0], oracle.sgm_census(small[0][0], small[0][1], 16, 5, 3, 60, 4))
    # and the first size again on top of the small call's bytes
    got, want = _both(sv, torch, big[:1], 64, 8, 1)
    assert np.array_equal(got, want)


def test_noalloc_limits_are_clear_errors():
    import torch
    import stereo_vision_amd as sv
    from stereo_vision_amd._lib import SgmxError
    H, W = 376, 1241
    L = torch.zeros((1, H, W), dtype=torch.uint8, device="cuda")
    out = torch.zeros((1, H, W), dtype=torch.int16, device="cuda")
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=128, window_sz=5, penalty1=3, penalty2=60))
    with pytest.raises(SgmxError) as ei:  # one KITTI pair at D=128 needs 480 MB
        e.compute_device_noalloc(L.data_ptr(), L.data_ptr(), out.data_ptr(), 1, W, H, 0)
    assert ei.value.code == 1 and "bytes of scratch" in str(ei.value)
    e.close()
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=16, window_sz=5, penalty1=3, penalty2=300))  # materialised formulation
    with pytest.raises(SgmxError) as ei:
        e.compute_device_noalloc(L.data_ptr(), L.data_ptr(), out.data_ptr(), 1, 64, 32, 0)
    assert ei.value.code == 1 and "volume-free" in str(ei.value)
    e.close()


def test_noalloc_under_stream_capture_replays_equal_dev(oracle):
    """The entry point's purpose: it may run where hipMalloc may not.  Two calls (the second with a smaller input and
    another lane layout) are captured into one HIP graph on a side stream in GLOBAL capture mode -- an allocation, a
    free or a synchronous copy anywhere in the process would invalidate the capture -- and the graph is replayed three
    times on fresh outputs: every replay equals sgmx_compute_census_u8_dev and the oracle."""
    import torch
    import stereo_vision_amd as sv
    big = [synth_pair(910 + i, 120, 200, 40) for i in range(2)]
    small = [synth_pair(920 + i, 61, 97, 12) for i in range(2)]
    eb = sv.StereoSGM(sv.StereoSGMParams(disp_range=64, window_sz=5, penalty1=3, penalty2=60, num_paths=8, max_batch=2))
    es = sv.StereoSGM(sv.StereoSGMParams(disp_range=16, window_sz=5, penalty1=3, penalty2=60, num_paths=4, max_batch=2))
    dev = lambda pairs, k: torch.from_numpy(np.stack([p[k] for p in pairs])).cuda()
    Lb, Rb, Ls, Rs = dev(big, 0), dev(big, 1), dev(small, 0), dev(small, 1)
    ob = torch.zeros((2, 120, 200), dtype=torch.int16, device="cuda")
    os_ = torch.zeros((2, 61, 97), dtype=torch.int16, device="cuda")
    wb, ws = torch.zeros_like(ob), torch.zeros_like(os_)
    st0 = torch.cuda.current_stream().cuda_stream
    for e, L, R, w_, (H, W) in ((eb, Lb, Rb, wb, (120, 200)), (es, Ls, Rs, ws, (61, 97))):
        assert e.lib.sgmx_compute_census_u8_dev(e.h, 2, L.data_ptr(), R.data_ptr(), W, H, w_.data_ptr(), st0) == 0
        # one eager call before the capture: the runtime loads a kernel's code object at its first launch
        assert e.lib.sgmx_compute_census_u8_dev_noalloc(e.h, 2, L.data_ptr(), R.data_ptr(), W, H, w_.data_ptr(), st0) == 0
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="global"):
        st = torch.cuda.current_stream().cuda_stream
        rc1 = eb.lib.sgmx_compute_census_u8_dev_noalloc(eb.h, 2, Lb.data_ptr(), Rb.data_ptr(), 200, 120, ob.data_ptr(), st)
        rc2 = es.lib.sgmx_compute_census_u8_dev_noalloc(es.h, 2, Ls.data_ptr(), Rs.data_ptr(), 97, 61, os_.data_ptr(), st)
    assert rc1 == 0, eb.lib.sgmx_last_error(eb.h)
    assert rc2 == 0, es.lib.sgmx_last_error(es.h)
    assert torch.cuda.mem_get_info()[0] >= free_before - (8 << 20)  # (the graph object itself; no scratch was allocated)
    for rep in range(3):
        ob.fill_(-1)
        os_.fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(ob, wb) and torch.equal(os_, ws), "replay %d" % rep
    got = ob.cpu().numpy().view(np.uint16)
    assert np.array_equal(got[1], oracle.sgm_census(big[1][0], big[1][1], 64, 5, 3, 60, 8))
    gots = os_.cpu().numpy().view(np.uint16)
    assert np.array_equal(gots[0], oracle.sgm_census(small[0][0], small[0][1], 16, 5, 3, 60, 4))
    del graph
    eb.close()
    es.close()
