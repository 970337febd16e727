"""One rank of a two-rank data-parallel training step through the HIP kernels (started by tests/test_gpu_dist2.py as
a fresh process; both ranks share GPU 0, the exchange goes through gloo -- RCCL refuses two ranks on one device).

    python tests/dist_worker.py <rank> <world> <port> <vanilla|ngp> <out.pt>

Every rank renders its contiguous share of one ray batch with replicated parameters; the gradients meet in
``GradBuckets`` (flat buffers, hook-issued all-reduce, the hash-grid model's two-phase backward) and every rank takes
the same ``FusedAdam`` step.  Rank 0 also runs the whole batch alone and writes both results for the test to compare.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lego_batch(dev, R, seed=0):
    from nerf_amd.Camera import Camera, look_at_camera
    eye = torch.tensor([3.2, 1.8, 1.7])
    c2w = look_at_camera(eye, 2 * eye, torch.tensor([0.0, 0.0, 1.0]))
    g = torch.Generator().manual_seed(seed)
    pix = torch.randperm(800 * 800, generator=g)[:R].to(torch.int32)
    cam = Camera(None, None, (1111.11, 1111.11), 800, 800, c2w=c2w.to(dev))
    o, d = cam.get_ray_batch(pix.to(dev))
    return o, d, torch.rand(R, 64, generator=g).to(dev), torch.rand(R, 128, generator=g).to(dev), torch.rand(R, 3, generator=g).to(dev)


def build(kind, dev):
    """(models, bucket lists, near, far, step keywords without the uniforms)"""
    from nerf_amd import render
    if kind == "ngp":
        from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
        torch.manual_seed(5)
        m = FastNGPNeRF().to(dev)
        with torch.no_grad():                       # the reference's init makes the outputs nearly constant: sharper weights
            gen = torch.Generator().manual_seed(6)
            for n, p in m.named_parameters():
                if n.startswith("hash_encoder"):
                    p.mul_(300.0)
                elif n != "fc_density.2.bias":
                    p.copy_((torch.randn(p.shape, generator=gen) * (0.5 if p.dim() > 1 else 0.1)).to(p.device))
        tables = list(m.hash_encoder.parameters())
        rest = [p for n, p in m.named_parameters() if not n.startswith("hash_encoder")]
        return [m], [tables, rest], 0.05, 1.2, dict(loss_fn=render.ngp_loss), 0.25
    from nerf_amd.models.NeRF import NeRF
    torch.manual_seed(0)
    coarse, fine = NeRF().to(dev), NeRF().to(dev)
    return [coarse, fine], [list(fine.parameters()), list(coarse.parameters())], 2.0, 6.0, \
        dict(white_bkgd=True, fine_model=fine, n_importance=128), 1.0


def main():
    rank, world, port, kind, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from nerf_amd import render
    from nerf_amd.dist import GradBuckets, shard_range
    from nerf_amd.optim import FusedAdam
    R = 8192
    o, d, u, uf, gt = lego_batch(dev, R)
    models, bucket_lists, near, far, kw, scale = build(kind, dev)
    o = o * scale
    named = [(("m%d." % i) + n, p) for i, m in enumerate(models) for n, p in m.named_parameters()]
    result = {}
    if rank == 0:   # the whole batch in one process, no buckets: the gradient every rank must end up with
        extra = dict(u_fine=uf) if kind == "vanilla" else {}
        for _, p in named:
            p.grad = None
        loss = render.train_step(models[0], o, d, gt, near, far, 64, t_rand=u, chunk_rays=R // world, **extra, **kw)
        result["full_loss"] = float(loss)
        result["full_grads"] = {n: p.grad.detach().cpu().clone() for n, p in named}
        for _, p in named:
            p.grad = None
    gb = GradBuckets(bucket_lists)
    assert gb.world == world
    opt = FusedAdam(gb.parameters(), lr=1e-2 if kind == "ngp" else 5e-4)
    lo, hi = shard_range(R, rank, world)
    extra = dict(u_fine=uf[lo:hi]) if kind == "vanilla" else {}
    order = []
    send = gb._send
    gb._send = lambda bi: (order.append(bi), send(bi))[1]
    loss = render.train_step(models[0], o[lo:hi], d[lo:hi], gt[lo:hi], near, far, 64, t_rand=u[lo:hi], buckets=gb,
                             **extra, **kw)
    torch.cuda.synchronize()
    part = torch.tensor([float(loss)], dtype=torch.float64)
    dist.all_reduce(part)                                         # mean over ranks of the per-rank mean losses
    result["dp_loss"] = float(part) / world
    result["send_order"] = order
    result["dp_grads"] = {n: p.grad.detach().cpu().clone() for n, p in named}
    opt.step()
    torch.cuda.synchronize()
    # every rank must hold the same parameters after the step: compare with rank 0's, bit for bit
    same = True
    for _, p in named:
        mine = p.detach().cpu()
        ref = mine.clone()
        dist.broadcast(ref, src=0)
        same = same and torch.equal(mine, ref)
    flag = torch.tensor([1.0 if same else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    result["params_identical_on_all_ranks"] = bool(flag.item() == 1.0)
    result["params_moved"] = True
    # inference: the image's rows split over the ranks and gathered (the path's only collective), against the whole
    # image rendered by one process.  37 rows: the shares are ragged.
    from nerf_amd.Camera import Camera, look_at_camera
    eye = torch.tensor([3.2, 1.8, 1.7]) * scale
    cam = Camera(None, None, (40.0, 41.0), 37, 24, c2w=look_at_camera(eye, 2 * eye, torch.tensor([0.0, 0.0, 1.0])).to(dev))
    img_o, img_d = cam.get_rays()
    ikw = dict(n_samples=32, return_depth=True)
    if kind == "vanilla":
        ikw.update(white_bkgd=True, fine_model=models[1], n_importance=48)
    img, dep = render.render_image_sharded(models[0], img_o, img_d, near, far, **ikw)
    if rank == 0:
        whole, whole_dep = render.render_image(models[0], img_o, img_d, near, far, **ikw)
        result["sharded_image_equal"] = bool(torch.equal(img, whole) and torch.equal(dep, whole_dep))
        result["sharded_image_shape"] = list(img.shape)
    if rank == 0:
        torch.save(result, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
