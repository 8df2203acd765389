"""Host logic of the training loop on CPU: the data-parallel gradient exchange with two gloo ranks,
the requires-grad part filters, option presets and lazy-regularisation bookkeeping.  (The networks
themselves need the GPU; their parity tests are in test_gpu_networks.py.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp
import torch.utils._python_dispatch


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, bucket_bytes, result_queue):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    from ga_gan_amd.training.training_loop import GradReducer
    torch.manual_seed(0)                      # same parameters on every rank
    params = [torch.nn.Parameter(torch.randn(s)) for s in ((300, 7), (5,), (64, 64), (3, 3, 3), ())]
    torch.manual_seed(100 + rank)             # different gradients per rank
    for i, p in enumerate(params):
        if i != 1:                            # one parameter received no gradient in this phase
            p.grad = torch.randn_like(p)
    local = [None if p.grad is None else p.grad.clone() for p in params]
    moved = GradReducer(world, bucket_bytes=bucket_bytes)([p for p in params if p.grad is not None])
    gathered = [None] * world
    torch.distributed.all_gather_object(gathered, local)
    ok = True
    for i, p in enumerate(params):
        if i == 1:
            ok &= p.grad is None
            continue
        mean = sum(g[i] for g in gathered) / world
        ok &= torch.allclose(p.grad, mean, atol=1e-6)
    expect_bytes = sum(p.numel() * 4 for i, p in enumerate(params) if i != 1)
    result_queue.put((rank, bool(ok), moved == expect_bytes))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('bucket_bytes', [64 << 20, 4096])
def test_grad_reducer_two_ranks_gloo(bucket_bytes):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, bucket_bytes, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True, True), (1, True, True)]


class _CountCopies(torch.utils._python_dispatch.TorchDispatchMode):
    """Counts the operators that would move gradient bytes around: concatenations and copies."""

    def __init__(self):
        super().__init__()
        self.seen = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if 'aten.cat' in name or 'aten.copy_' in name or 'aten.clone' in name or 'aten._foreach_copy' in name:
            self.seen.append(name)
        return func(*args, **(kwargs or {}))


def _rank_overlap(rank, world, port, bucket_bytes, result_queue):
    """Two backward calls per phase (as Dmain: generated, then real images).  The gradients accumulate in the
    reducer's buckets (prepare); it is armed before the second call, enqueues a bucket's all-reduce from the
    post-accumulate hook of the bucket's last member, and must end with the rank average of the accumulated
    gradients - with one collective per bucket and without packing or unpacking copies."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    from ga_gan_amd.training.training_loop import GradReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(20, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, 3))
    frozen = torch.nn.Linear(3, 3).requires_grad_(False)   # attached while frozen, takes no part in the phase
    unused = torch.nn.Parameter(torch.ones(5))          # trainable, prepared, but no backward call reaches it
    net.requires_grad_(False)                           # (the trainer attaches to frozen networks and enables parts per phase)
    extra = torch.nn.Parameter(torch.ones(4))           # not attached, not prepared: a loose gradient, covered by finish()
    reducer = GradReducer(world, bucket_bytes=bucket_bytes)
    reducer.attach(list(net.parameters()) + list(frozen.parameters()) + [unused])
    net.requires_grad_(True)
    torch.manual_seed(100 + rank)
    x1, x2 = torch.randn(8, 20), torch.randn(8, 20)
    # this rank's own gradients, from a copy of the network that the reducer does not know (the buckets are
    # all-reduced where they lie, so after the armed backward call `.grad` may already hold the sum over ranks)
    import copy
    twin = copy.deepcopy(net)
    (twin(x1).square().mean() + twin(x2).square().mean()).backward()
    own = [q.grad.clone() for q in twin.parameters()] + [torch.full([4], float(rank + 1))]
    ok = True
    trainable = list(net.parameters()) + [unused]
    nbytes = sum(q.numel() * 4 for q in trainable)
    for phase in range(2):                              # hooks and buckets stay from phase to phase
        extra.grad = None
        reducer.prepare(trainable)
        views = [q.grad for q in trainable]
        ok &= all(float(v.abs().sum()) == 0 for v in views)
        net(x1).square().mean().backward()              # not armed: nothing is exchanged yet
        ok &= len(reducer._pending) == 0
        ok &= all(q.grad.data_ptr() == v.data_ptr() for q, v in zip(trainable, views))     # accumulated in place
        local_first = [q.grad.clone() for q in net.parameters()]
        before = reducer.collectives
        with _CountCopies() as copies:
            reducer.arm()
            (net(x2).square().mean() + (extra * (rank + 1)).sum()).backward()
            in_flight = len(reducer._pending)           # buckets already handed to the backend during backward
            moved = reducer.finish(trainable + [extra])
        params = list(net.parameters()) + [extra]
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, own)
        for i, q in enumerate(params):
            mean = sum(g[i] for g in gathered) / world
            ok &= torch.allclose(q.grad, mean, atol=1e-5)
        ok &= unused.grad is None                       # as if it had never been given a gradient
        ok &= moved == nbytes + extra.numel() * 4
        ok &= all(a.abs().sum() > 0 for a in local_first)
        buckets = len(reducer._plans[tuple(id(q) for q in trainable)].buffers)
        ok &= buckets == (1 if bucket_bytes >= nbytes else buckets) and buckets <= -(-nbytes // bucket_bytes) + len(trainable)
        ok &= reducer.collectives - before == buckets + 1            # + 1: the loose gradient of `extra`
        # the only copies: packing / unpacking the ONE loose gradient (cat + copy_)
        ok &= len(copies.seen) == 2
        if bucket_bytes < 1 << 20:
            ok &= buckets >= 3 and in_flight >= 2       # small buckets: the exchange started before backward ended
    ok &= all(q.grad is None for q in frozen.parameters())
    result_queue.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('bucket_bytes', [64 << 20, 8192])
def test_grad_reducer_overlapped_with_backward_two_ranks_gloo(bucket_bytes):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_overlap, args=(r, 2, port, bucket_bytes, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_grad_reducer_bucket_plan():
    """Buckets hold whole tensors in reverse parameter order, none larger than the limit unless a single tensor is."""
    from ga_gan_amd.training.training_loop import GradReducer
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (1000, 10, 3000, 500, 500, 20)]
    plan = GradReducer._Plan(params, bucket_bytes=4096)
    sizes = [b.numel() for b in plan.buffers]
    assert sizes == [20 + 500 + 500, 3000, 10 + 1000]          # from the last parameter backwards
    assert plan.bucket_of[id(params[5])] == 0 and plan.bucket_of[id(params[2])] == 1 and plan.bucket_of[id(params[0])] == 2
    for p in params:
        v = plan.views[id(p)]
        assert v.shape == p.shape and v.is_contiguous()
    a, b = plan.views[id(params[4])], plan.views[id(params[3])]
    assert b.data_ptr() == a.data_ptr() + 500 * 4              # packed back to back


def test_grad_reducer_single_rank_is_a_no_op():
    from ga_gan_amd.training.training_loop import GradReducer
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    assert GradReducer(1)([p]) == 0
    assert torch.equal(p.grad, torch.full((3,), 2.0))


def test_requires_grad_parts_match_reference_counts():
    """Trainable-parameter counts of the few-shot recipes at FFHQ-256 (paper256 channels) and
    FFHQ-1024: 4,005,888 (Affine+ @256) and 637,824 (AffineLight+ @1024)."""
    from ga_gan_amd.training import networks, training_loop
    opts = training_loop.setup_training_options(cfg='paper256', resolution=256, batch_size=4, **training_loop.AFFINE_PLUS)
    G = networks.Generator(**opts['G_kwargs'])
    training_loop.set_requires_grad(G, opts['requires_grad_parts']['Gmain'])
    assert sum(p.numel() for p in G.parameters() if p.requires_grad) == 4005888
    assert sum(p.numel() for p in G.parameters()) == 24964834
    D = networks.Discriminator(**opts['D_kwargs'])
    assert sum(p.numel() for p in D.parameters()) == 24001089
    assert opts['D_kwargs']['epilogue_kwargs']['mbstd_group_size'] == 4      # batch 4 < 8
    opts = training_loop.setup_training_options(cfg='stylegan2', resolution=1024, batch_size=16, **training_loop.AFFINE_LIGHT_PLUS)
    G = networks.Generator(**opts['G_kwargs'])
    training_loop.set_requires_grad(G, opts['requires_grad_parts']['Gmain'])
    assert sum(p.numel() for p in G.parameters() if p.requires_grad) == 637824
    training_loop.set_requires_grad(G, ['all'])
    assert all(p.requires_grad for p in G.parameters())


def test_state_dict_names_match_reference_fixture():
    """The product modules expose exactly the reference's parameter / buffer names and shapes."""
    import golden_io
    from ga_gan_amd.training import networks
    for case in golden_io.load('networks'):
        m = case.meta
        want = {k: tuple(v.shape) for k, v in case.group('sd.').items()}
        if m['kind'] == 'G':
            sk = dict(channel_base=m['channel_base'], channel_max=m['channel_max'], num_fp16_res=0,
                      conv_clamp=m['conv_clamp'], generator_requires_grad_parts=list(m['parts']))
            if m['parametrization'] is not None:
                sk.update(use_domain_modulation=True, domain_modulation_parametrization=m['parametrization'])
            net = networks.Generator(z_dim=m['z_dim'], c_dim=m['c_dim'], w_dim=m['w_dim'], img_resolution=m['res'],
                                     img_channels=3, mapping_kwargs=dict(num_layers=m['mapping_layers']), synthesis_kwargs=sk)
        else:
            extra = dict(block_kwargs=dict(freeze_layers=3)) if m['tag'] == 'D32_freeze' else {}
            net = networks.Discriminator(c_dim=m['c_dim'], img_resolution=m['res'], img_channels=3,
                                         architecture=m['architecture'], channel_base=m['channel_base'],
                                         channel_max=m['channel_max'], conv_clamp=m['conv_clamp'],
                                         epilogue_kwargs=dict(mbstd_group_size=m['mbstd_group_size']), **extra)
        got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert got == want, m['tag']


def test_conv_plans_tap_bookkeeping():
    from ga_gan_amd import conv2d_plugin as cp
    plan = cp.plan_conv_transpose(3, 3, stride=2, padding=0)
    assert plan.out_size(16, 16) == (33, 33) and plan.out_step == 2 and len(plan.phases) == 4
    assert sum(ph[4] * ph[5] for ph in plan.phases) == 9          # every tap belongs to exactly one phase
    w = torch.arange(2 * 3 * 9, dtype=torch.float32).reshape(2, 3, 3, 3)      # [ci, co, kh, kw]
    assert plan.pack(w, transposed_layout=True).numel() == w.numel()
    assert cp.plan_conv(3, 3, stride=2).out_size(33, 33) == (16, 16)
    assert cp.plan_conv(1, 1).pack(torch.ones(4, 5, 1, 1)).shape == (20,)
