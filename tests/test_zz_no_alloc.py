"""rag_merge_topk_device_static (the merge that allocates nothing: its keys live in a static
device array of the library) against rag_merge_topk_device, through the C ABI: bit-identical
outputs on two calls in a row, the second smaller than the first, plus the size limit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lists(rng, nlists, nq, k, torch):
    """nlists result sets [nq][k], each descending with distinct ids, some slots unfilled."""
    D = np.sort(rng.standard_normal((nlists, nq, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    I = rng.permutation(nlists * nq * k).astype(np.int64).reshape(nlists, nq, k)
    D[0, 0, k // 2:] = -np.finfo(np.float32).max           # an under-filled list
    I[0, 0, k // 2:] = -1
    if nlists > 1:
        D[1, :, 0] = D[0, :, 0]                             # ties between lists
    return torch.from_numpy(D).cuda(), torch.from_numpy(I).cuda()


def _call(fn, D, I, torch):
    nl, nq, k = D.shape
    Do = torch.full((nq, k), 7.0, dtype=torch.float32, device="cuda")
    Io = torch.full((nq, k), 7, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    rc = fn(nl, nq, k, C.c_void_p(D.data_ptr()), 0, C.c_void_p(I.data_ptr()), 0, C.c_void_p(Do.data_ptr()),
            C.c_void_p(Io.data_ptr()), 0, C.c_void_p(stream))
    torch.cuda.synchronize()
    return rc, Do.cpu().numpy(), Io.cpu().numpy()


def test_static_merge_equals_allocating_merge_two_calls_in_a_row():
    import torch
    from rag_amd import _lib
    lib = _lib.lib
    rng = np.random.default_rng(3)
    # first call large (k > 64 takes the LDS merge kernel), second smaller (register merge kernel):
    # the second must not see what the first left in the array
    for nlists, nq, k in ((8, 500, 100), (3, 7, 10), (8, 4096, 10), (2, 1, 64)):
        D, I = _lists(rng, nlists, nq, k, torch)
        rc_s, Ds, Is = _call(lib.rag_merge_topk_device_static, D, I, torch)
        assert rc_s == 0, _lib.last_error()
        rc_a, Da, Ia = _call(lib.rag_merge_topk_device, D, I, torch)
        assert rc_a == 0, _lib.last_error()
        assert np.array_equal(Ds, Da) and np.array_equal(Is, Ia), (nlists, nq, k)
        # and both are the right answer: best first, ties to the lower list
        Dc = D.cpu().numpy().transpose(1, 0, 2).reshape(nq, nlists * k)
        Ic = I.cpu().numpy().transpose(1, 0, 2).reshape(nq, nlists * k)
        slot = np.broadcast_to(np.arange(nlists * k), Dc.shape)
        valid = Ic >= 0
        order = np.lexsort((slot, -Dc.astype(np.float64), ~valid), axis=1)[:, :k]
        want_I = np.take_along_axis(Ic, order, 1)
        want_D = np.take_along_axis(Dc, order, 1)
        assert np.array_equal(Is, want_I) and np.array_equal(Ds, want_D), (nlists, nq, k)


def test_static_merge_rejects_what_does_not_fit():
    import torch
    from rag_amd import _lib
    lib = _lib.lib
    nlists, nq, k = 8, 4097, 32                     # 1,048,832 keys > 2^20
    D = torch.zeros((nlists, nq, k), dtype=torch.float32, device="cuda")
    I = torch.zeros((nlists, nq, k), dtype=torch.int64, device="cuda")
    rc, _, _ = _call(lib.rag_merge_topk_device_static, D, I, torch)
    assert rc == _lib.RAG_ERR_INVALID and "static scratch" in _lib.last_error()
    rc, _, _ = _call(lib.rag_merge_topk_device, D, I, torch)
    assert rc == 0
