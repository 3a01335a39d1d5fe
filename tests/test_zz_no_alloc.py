"""pt_trace_closest_noalloc_dev (closest hits with every intermediate in the library's built-in device block)
against pt_trace_closest_fast_dev (the same kernels over a pool the library allocates), through the C-ABI with
device pointers. Bar: bit-exact triangle ids and t. Two calls in a row share the block; the second is smaller, so
stale contents of the first would show."""
import numpy as np
import pytest

import ptamd
from ptamd import scenes, lib
from ptamd.types import RAY_DTYPE, HIT_DTYPE
import util

pytestmark = pytest.mark.gpu


def _device_rays(torch, rays):
    host = torch.from_numpy(np.frombuffer(np.ascontiguousarray(rays).tobytes(), dtype=np.uint8).copy())
    return host.cuda()


def _hits(torch, buf, n):
    torch.cuda.synchronize()
    return np.frombuffer(buf.cpu().numpy().tobytes(), dtype=HIT_DTYPE)[:n]


def test_noalloc_closest_hits_equal_the_allocating_entry_point():
    import torch
    b = util.atrium(detail=0.35)
    P = scenes.atrium_params(ptamd.default_params(64, 64))
    r = b.renderer(P)
    n_max = lib.PT_NOALLOC_MAX_RAYS
    rays = scenes.random_rays(n_max, seed=33, extent=1400.0)
    rays['o'][:, 1] = np.abs(rays['o'][:, 1]) * 0.8
    rays['d'][5] = [np.nan, 0, 1]
    rays['o'][6] = [np.inf, 0, 0]
    d_rays = _device_rays(torch, rays)
    want_buf = torch.zeros(n_max * HIT_DTYPE.itemsize, dtype=torch.uint8, device='cuda')
    got_buf = torch.full((n_max * HIT_DTYPE.itemsize,), 0xAB, dtype=torch.uint8, device='cuda')
    # first call: the largest batch the block takes; second call: a smaller, ragged one over the same block
    for n in (n_max, 70001, 1):
        want_buf.zero_(); got_buf.fill_(0xAB)
        torch.cuda.synchronize()
        r.trace_closest_fast_dev(d_rays.data_ptr(), n, want_buf.data_ptr())
        r.sync()
        want = _hits(torch, want_buf, n_max).copy()
        r.trace_closest_noalloc_dev(d_rays.data_ptr(), n, got_buf.data_ptr())
        r.sync()
        got = _hits(torch, got_buf, n_max).copy()
        assert np.array_equal(got['tri'][:n], want['tri'][:n]), n
        assert np.array_equal(got['t'][:n].view(np.uint32), want['t'][:n].view(np.uint32)), n
        assert (got['tri'][:n] != 0xFFFFFFFF).sum() > 0 or n == 1
        # nothing past the batch is written
        assert np.all(np.frombuffer(got.tobytes(), np.uint8)[n * HIT_DTYPE.itemsize:] == 0xAB), n
    # the property the entry point is named for: on a warmed context a call leaves the device's free memory where it
    # was (a hipMalloc anywhere on the way, even one freed again later by a caching allocator, would show here or in
    # the driver's own count below)
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    reserved_before = torch.cuda.memory_reserved()
    for n in (n_max, 4097, n_max):
        r.trace_closest_noalloc_dev(d_rays.data_ptr(), n, got_buf.data_ptr())
    r.sync()
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after == free_before, (free_before, free_after)
    assert torch.cuda.memory_reserved() == reserved_before
    # (the allocating entry point, by contrast, grows the library's pool when the batch does)
    # a second context on the same GPU shares the block; calls in sequence do not disturb each other
    r2 = b.renderer(P)
    r2.trace_closest_noalloc_dev(d_rays.data_ptr(), 5000, got_buf.data_ptr())
    r2.sync()
    r.trace_closest_fast_dev(d_rays.data_ptr(), 5000, want_buf.data_ptr())
    r.sync()
    got, want = _hits(torch, got_buf, 5000), _hits(torch, want_buf, 5000)
    assert np.array_equal(got['tri'], want['tri']) and np.array_equal(got['t'].view(np.uint32), want['t'].view(np.uint32))
    r2.close()
    # sizes beyond the block are refused with a message, not truncated
    with pytest.raises(ptamd.PtError, match='exceed'):
        r.trace_closest_noalloc_dev(d_rays.data_ptr(), n_max + 1, got_buf.data_ptr())
    with pytest.raises(ptamd.PtError):
        r.trace_closest_noalloc_dev(d_rays.data_ptr(), 0, got_buf.data_ptr())
    r.close()
