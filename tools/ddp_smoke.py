"""Two data-parallel ranks of the trainer on ONE GPU (gloo moves the gradient buckets through the host): checks
that the overlapped gradient exchange leaves both ranks with identical parameters after a few iterations, and
that they differ from a single-rank run only by the batch they saw.  A rehearsal of the multi-GPU path where
only one device is available; the real thing runs over RCCL (bench.py --gpus N)."""
import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp


def rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from ga_gan_amd.training import training_loop
    opts = training_loop.setup_training_options(cfg='paper256', resolution=64, batch_size=8, num_gpus=world, batch_gpu=4,
                                                glrate=0.02, **training_loop.AFFINE_PLUS)
    for k in ('G_kwargs', 'D_kwargs'):
        kw = opts[k].get('synthesis_kwargs', opts[k])
        kw['channel_base'], kw['channel_max'] = 2048, 64
    trainer = training_loop.Trainer(rank=rank, random_seed=0, **opts)
    if os.environ.get('DDP_SMOKE_AFTER'):     # exchange after the backward pass instead of from the hooks
        trainer.loss.grad_sync = None
    print(f'rank {rank}: trainer ready', flush=True)
    gen = torch.Generator(device='cuda').manual_seed(1 + rank)
    for _ in range(5):     # covers Gmain, Greg, Dmain (two backward calls, armed before the second); Dreg needs 16
        img, c = trainer.synthetic_batch(gen)
        trainer.step(img, c)
        torch.cuda.synchronize()
        print(f'rank {rank}: step done, {trainer.exchanged_bytes} bytes exchanged so far', flush=True)
    torch.cuda.synchronize()
    digest = {n: p.detach().double().sum().item() for n, p in list(trainer.G.named_parameters()) + list(trainer.D.named_parameters())}
    q.put((rank, digest, trainer.exchanged_bytes))
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    res = []
    while len(res) < len(procs):     # a rank that died never reports: fail at once instead of waiting for it
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead:
                for p in procs:
                    p.kill()
                raise SystemExit(f'a rank exited with {dead}')
    res.sort(key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    worst = max(abs(res[0][1][n] - res[1][1][n]) / max(abs(res[0][1][n]), 1e-6) for n in res[0][1])
    print(f'ranks agree to {worst:.2e} (relative, per-parameter sums); bytes exchanged per rank: {res[0][2]}')
    assert worst < 1e-6 and res[0][2] > 0 and res[0][2] == res[1][2]
    print('OK')
