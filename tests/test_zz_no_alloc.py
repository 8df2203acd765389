"""The entry point that allocates nothing (`sg_conv2d_reduced`) against `sg_conv2d` with a caller-provided
workspace, through the C ABI, on two calls in a row (the second smaller, so stale partial sums of the first
would show).  The convolution output comes from the same kernel and must be bit-identical; the reduced sums are
the same float32 partial sums added in a different order: 1e-5 of the largest sum.  Run with -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _run(cp, n, ci, co, h, w, dtype, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn([n, ci, h, w], device=DEV, generator=g).to(dtype)
    wt = (torch.randn([co, ci, 3, 3], device=DEV, generator=g) / (ci * 9) ** 0.5).to(dtype)
    s = torch.rand([n, co], device=DEV, generator=g) + 0.5
    rx = torch.randn([n, co, h, w], device=DEV, generator=g).to(dtype)
    plan = cp.plan_conv(3, 3, padding=1)
    packed = plan.pack(wt)
    y0, r0 = cp.conv2d(x, packed, plan, co, out_scale=s, reduce_x=rx)
    out = torch.full_like(y0, float('nan'))
    rsum = torch.full([n, co], float('nan'), device=DEV)
    need = cp.workspace_bytes(x, plan, co)               # float32: the scratch of the half-split kernels is the caller's
    assert (need > 0) == (dtype == torch.float32)
    work = torch.empty([need], dtype=torch.uint8, device=DEV) if need else None
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()['allocation.all.allocated']
    y1, r1 = cp.conv2d(x, packed, plan, co, out_scale=s, reduce_x=rx, out=out, reduce_sum=rsum, workspace=work)
    assert torch.cuda.memory_stats()['allocation.all.allocated'] == before, 'the call allocated device memory through torch'
    assert y1 is out and r1 is rsum
    assert torch.equal(y1, y0)
    scale = float(r0.abs().max())
    assert float((r1 - r0).abs().max()) <= 1e-5 * scale, (float((r1 - r0).abs().max()), scale)
    # and against float64 arithmetic on the host: sum over pixels of reduce_x * conv(x, w)
    want = (torch.nn.functional.conv2d(x.double().cpu(), wt.double().cpu(), padding=1) * rx.double().cpu()).sum(dim=(2, 3))
    assert float((r1.double().cpu() - want).abs().max()) <= 2e-4 * float(want.abs().max())


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_conv2d_reduced_twice_second_smaller(dtype):
    from ga_gan_amd import conv2d_plugin as cp
    _run(cp, 3, 64, 64, 48, 64, dtype, 1)     # 3 x 12 rows x 64 channels of partial sums
    _run(cp, 2, 32, 32, 16, 32, dtype, 2)     # fewer samples, rows and channels: must not see the first call's sums
    _run(cp, 5, 64, 32, 4, 4, dtype, 3)       # several samples per tile: the kernel adds into the array with atomics


def test_conv2d_reduced_rejects_oversize_and_bad_arguments():
    import ctypes
    from ga_gan_amd import _lib, conv2d_plugin as cp
    plan = cp.plan_conv(3, 3, padding=1)
    n, c, h = 16, 128, 512                    # 16 x (64 x 16) rows x 128 channels = 2^21 partial sums > 2^20
    x = torch.zeros([1], device=DEV)
    p = cp.Conv2dParams(x=_lib.ptr(x), w=_lib.ptr(x), y=_lib.ptr(x), reduce_x=_lib.ptr(x), dtype=2, n=n, ci=c, h=h, w_in=h,
                        co=c, oh=h, ow=h, in_step=1, out_step=1, num_phases=1, act=1, gain=1.0, clamp=-1.0)
    p.phase[0] = cp.Phase(0, 0, -1, -1, 3, 3, 0)
    fn = _lib.lib().sg_conv2d_reduced
    assert fn(ctypes.byref(p), _lib.ptr(x), None) == -2
    assert b'exceed the fixed array' in _lib.lib().sg_last_error()
    assert fn(ctypes.byref(p), None, None) == -3
    p.reduce_out = _lib.ptr(x)
    assert fn(ctypes.byref(p), _lib.ptr(x), None) == -1
    p.reduce_out = None
    p.dtype, p.n, p.h, p.w_in, p.oh, p.ow = 0, 1, 16, 16, 16, 16     # float32 without a workspace: refused, not allocated for
    assert fn(ctypes.byref(p), _lib.ptr(x), None) == -1
    assert b'allocates nothing' in _lib.lib().sg_last_error()


def test_conv2d_reduced_replays_from_a_graph():
    """Recorded into a HIP graph (where hipMalloc fails and a memset node would act once), the call must clear its
    array on every replay: the geometry whose kernel adds into the array with atomics, replayed three times on new
    inputs, against sg_conv2d with a workspace each time."""
    from ga_gan_amd import _lib, conv2d_plugin as cp
    assert _lib.lib().sg_init() == 0          # resolves the array's address: not allowed inside a capture
    n, ci, co, h, w = 5, 64, 32, 4, 4
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn([n, ci, h, w], device=DEV, generator=g).to(torch.bfloat16)
    wt = (torch.randn([co, ci, 3, 3], device=DEV, generator=g) / (ci * 9) ** 0.5).to(torch.bfloat16)
    rx = torch.randn([n, co, h, w], device=DEV, generator=g).to(torch.bfloat16)
    plan = cp.plan_conv(3, 3, padding=1)
    packed = plan.pack(wt)
    out = torch.empty([n, co, h, w], device=DEV, dtype=torch.bfloat16)
    rsum = torch.empty([n, co], device=DEV)
    free_before = torch.cuda.mem_get_info()[0]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cp.conv2d(x, packed, plan, co, reduce_x=rx, out=out, reduce_sum=rsum)
    for trial in range(3):
        x.copy_(torch.randn([n, ci, h, w], device=DEV, generator=g).to(torch.bfloat16))
        rsum.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        y0, r0 = cp.conv2d(x, packed, plan, co, reduce_x=rx)
        assert torch.equal(out, y0), trial
        scale = float(r0.abs().max())
        assert float((rsum - r0).abs().max()) <= 1e-5 * scale, (trial, float((rsum - r0).abs().max()), scale)
    del graph
    # nothing was allocated behind torch's back either (the driver's own view of the device's free memory; torch's
    # allocator may have cached blocks of its own, which only lowers the figure by what the test itself asked for)
    assert free_before - torch.cuda.mem_get_info()[0] < (64 << 20)
