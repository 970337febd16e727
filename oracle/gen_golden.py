"""Generate tests/golden/*.npz by importing the reference (skowae/NeRF) itself.

Run only in the development container, where the reference checkout exists:

    MPLBACKEND=Agg python oracle/gen_golden.py [/root/reference [g1 g2 ... g10]]

The outputs are small input/output vectors of the reference's own functions
(plus per-parameter checksums proving that seeded construction gives the same
weights).  Nothing of the reference's source travels: only numbers.
TEST INFRASTRUCTURE ONLY.
"""
import json
import os
import sys

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
os.environ.setdefault("MPLBACKEND", "Agg")

from Camera import Camera, look_at_camera  # noqa: E402  (reference)
from models.NeRF import NeRF  # noqa: E402
from models.FastMLPNeRF import FastMLPNeRF  # noqa: E402
from models.FastNGPNeRF import FastNGPNeRF  # noqa: E402
from utils.positional_encoding import PositionalEncoding  # noqa: E402
from utils.hash_grid_encoder import HashGridEncoder  # noqa: E402
import utils.get_rays as legacy  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                    for k, v in arrs.items()})
    print("wrote", name, len(arrs), "arrays")


def param_checksums(model):
    return {k: np.array([float(v.double().sum()), float(v.double().abs().sum())])
            for k, v in model.state_dict().items()}


def sample_idx(n, k, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n, (min(k, n),), generator=g)


# ---------------------------------------------------------------- G1 ray generation
def g1():
    with open(os.path.join(REF, "src/data/skow/robot/transforms.json")) as f:
        meta = json.load(f)
    frames = meta["frames"]
    poses = [torch.from_numpy(np.array(fr["transform_matrix"])).float() for fr in frames]
    centers = torch.stack([p[:3, 3] for p in poses])
    centre = centers.mean(0)
    scale = 1.2 * (centers - centre).norm(dim=-1).max()
    pose = poses[0].clone()
    pose[:3, 3] = (pose[:3, 3] - centre) / scale
    fr = frames[0]
    W, H = 100, 75
    sw, sh = W / fr["w"], H / fr["h"]
    focal = (fr["fl_x"] * sw, fr["fl_y"] * sh)
    pp = (fr["cx"] * sw, fr["cy"] * sh)
    cam = Camera(eye=pose[:, 3], target=None, focal=focal, H=H, W=W, c2w=pose, pp=pp)
    ro, rd = cam.get_rays()
    out = dict(robot_c2w=pose[:3, :4].contiguous(), robot_intr=np.array([focal[0], focal[1], pp[0], pp[1]], dtype=np.float64),
               robot_hw=np.array([H, W]), robot_o=ro.contiguous(), robot_d=rd)
    # look-at camera (default principal point)
    eye = torch.tensor([1.0, 1.0, 2.0])
    tgt = torch.zeros(3)
    cam = Camera(eye=eye, target=tgt, focal=(100.0, 100.0), H=100, W=100)
    ro, rd = cam.get_rays()
    out.update(look_c2w=cam.c2w, look_o=ro.contiguous(), look_d=rd,
               look_up_z=look_at_camera(torch.tensor([0.0, 3.0, 0.0]), tgt))
    # LLFF pose 0 (stored [3,5]: 3x4 pose + hwf column)
    pb = np.load(os.path.join(REF, "src/data/llff/testscene/poses_bounds.npy"))
    p0 = pb[0, :15].reshape(3, 5)
    c2w = torch.from_numpy(p0[:, :4]).float()
    f = float(p0[2, 4]) / 8.0
    H, W = 378 // 6, 504 // 6
    cam = Camera(eye=c2w[:, 3], target=None, focal=(f / 6, f / 6), H=H, W=W, c2w=c2w)
    ro, rd = cam.get_rays()
    out.update(llff_c2w=c2w, llff_f=np.array([f / 6]), llff_hw=np.array([H, W]), llff_o=ro.contiguous(), llff_d=rd,
               llff_bounds=pb[0, 15:])
    save("g1_raygen.npz", **out)
    return out


# ---------------------------------------------------------------- G2 stratified sampling
def g2(g1o):
    ro = g1o["look_o"][10:14, 20:28].contiguous()
    rd = g1o["look_d"][10:14, 20:28].contiguous()
    out = dict(rays_o=ro, rays_d=rd)
    for tag, (near, far, n) in {"a": (2.0, 6.0, 64), "b": (0.05, 1.2, 64), "c": (0.25, 0.75, 17)}.items():
        torch.manual_seed(1)
        pts, z = Camera.sample_points_along_rays(ro, rd, near, far, n, perturb=True)
        torch.manual_seed(1)
        u = torch.rand_like(z)  # the same draw the call above made
        pts0, z0 = Camera.sample_points_along_rays(ro, rd, near, far, n, perturb=False)
        out.update({f"{tag}_cfg": np.array([near, far, n]), f"{tag}_u": u, f"{tag}_z": z, f"{tag}_pts": pts,
                    f"{tag}_z0": z0.contiguous(), f"{tag}_pts0": pts0, f"{tag}_lin": torch.linspace(near, far, n)})
    save("g2_stratified.npz", **out)
    return out


# ---------------------------------------------------------------- G3 positional encoding
def g3():
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(64, 3, generator=g) * 12 - 6)
    x[0] = 0.0
    x[1] = torch.tensor([6.0, -6.0, 6.0])
    x[2] = torch.tensor([1e-8, -1e-8, 3.14159265])
    save("g3_pe.npz", x=x, pe10=PositionalEncoding(10)(x), pe4=PositionalEncoding(4)(x))


# ---------------------------------------------------------------- G4 hash grid
def fill_tables(enc):
    """Closed-form table contents (level l, row r, feature f) so the far side
    can rebuild them without shipping 41 MB."""
    with torch.no_grad():
        for l, emb in enumerate(enc.embeddings):
            r = torch.arange(emb.weight.shape[0], dtype=torch.float64)
            for f in range(2):
                emb.weight[:, f] = (torch.sin(r * (0.37 + 0.011 * l) + 1.3 * f + 0.5 * l) * 0.5).float()


def g4():
    enc = HashGridEncoder()
    fill_tables(enc)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(512, 3, generator=g) * 2 - 1
    x[0] = 0.0
    x[1] = 1.0
    x[2] = -1.0
    x[3] = torch.tensor([0.5, -0.5, 0.25])        # exact cell boundaries on pow2 levels
    x[4] = torch.tensor([-1e-7, 1e-7, -0.999999])
    x[5] = torch.tensor([1.5, -2.25, 3.0])        # outside the unit cube
    x[6] = torch.tensor([1.0 / 16, 3.0 / 32, -5.0 / 64])
    x.requires_grad_(False)
    feats = enc(x)
    idx = []
    wts = []
    for emb, res in zip(enc.embeddings, enc.resolutions):
        size = min(enc.hash_size, res ** 3)
        pos = x.unsqueeze(1) * res
        p0 = pos.floor().long()
        w = pos - p0.float()
        p1 = (p0 + 1) % res
        p0 = p0 % res
        lv = []
        for bx in (0, 1):
            for by in (0, 1):
                for bz in (0, 1):
                    c = torch.cat([(p1 if bx else p0)[..., 0:1], (p1 if by else p0)[..., 1:2],
                                   (p1 if bz else p0)[..., 2:3]], dim=-1)
                    lv.append(enc.hash_fn(c, size).squeeze(1))
        idx.append(torch.stack(lv))
        wts.append(w.squeeze(1))
    idx = torch.stack(idx).to(torch.int32)  # [16,8,512]
    # backward: d(sum(feats * a)) / d tables
    a = torch.rand(512, 32, generator=g) - 0.5
    (feats * a).sum().backward()
    gsum = np.array([float(e.weight.grad.double().sum()) for e in enc.embeddings])
    gl2 = np.array([float(e.weight.grad.double().norm()) for e in enc.embeddings])
    rows = [sample_idx(e.weight.shape[0], 64, 40 + l) for l, e in enumerate(enc.embeddings)]
    # sample rows that are actually touched: take them from the corner indices
    rows = [idx[l].reshape(-1)[sample_idx(8 * 512, 64, 40 + l)].long() for l in range(16)]
    grows = torch.stack([e.weight.grad[r] for e, r in zip(enc.embeddings, rows)])
    save("g4_hash.npz", x=x, feats=feats.detach(), idx=idx, w=torch.stack(wts), a=a,
         resolutions=np.array(enc.resolutions), sizes=np.array([e.weight.shape[0] for e in enc.embeddings]),
         grad_sum=gsum, grad_l2=gl2, grad_rows=torch.stack(rows), grad_vals=grows)


# ---------------------------------------------------------------- G5 models
def model_case(name, model, x, d, g, boost=None):
    if boost:
        with torch.no_grad():
            for k, p in model.named_parameters():
                for pat, s in boost.items():
                    if k.startswith(pat):
                        p.mul_(s)
    rgb, sigma = model(x, d)
    a = torch.rand(rgb.shape, generator=g) - 0.5
    b = torch.rand(sigma.shape, generator=g) - 0.5
    loss = (rgb * a).sum() + (sigma * b).sum()
    model.zero_grad()
    loss.backward()
    out = {f"{name}_rgb": rgb.detach(), f"{name}_sigma": sigma.detach(), f"{name}_a": a, f"{name}_b": b}
    for k, p in model.named_parameters():
        gr = p.grad.reshape(-1)
        ii = sample_idx(gr.numel(), 64, 7)
        out[f"{name}_gnorm/{k}"] = np.array([float(gr.double().norm())])
        out[f"{name}_gidx/{k}"] = ii
        out[f"{name}_gval/{k}"] = gr[ii]
    return out


def g5():
    g = torch.Generator().manual_seed(5)
    x = torch.rand(256, 3, generator=g) * 8 - 4
    d = torch.randn(256, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    out = dict(x=x, d=d)
    torch.manual_seed(0)
    m = NeRF()
    for k, v in param_checksums(m).items():
        out[f"nerf_ck/{k}"] = v
    out.update(model_case("nerf", m, x, d, g))
    torch.manual_seed(0)
    m = FastMLPNeRF()
    for k, v in param_checksums(m).items():
        out[f"fastmlp_ck/{k}"] = v
    out.update(model_case("fastmlp", m, x, d, g))
    save("g5_mlp.npz", **out)

    xn = torch.rand(256, 3, generator=g) * 2 - 1
    out = dict(x=xn, d=d * 1.7)  # un-normalised dirs: the model normalises them itself
    torch.manual_seed(0)
    m = FastNGPNeRF()
    for k, v in param_checksums(m).items():
        out[f"ngp_ck/{k}"] = v
    out.update(model_case("ngp", m, xn, d * 1.7, g))
    out["ngp_density"] = m.query_density(xn)
    rgb0, sig0 = m(xn[:16], None)
    out["ngp_nodir_rgb"] = rgb0.detach()
    boost = {"hash_encoder": 1e3, "fc_density.0.weight": 1e4, "fc_density.2.weight": 3e3,
             "fc_color.0.weight": 1e4, "fc_color.2.weight": 1e4, "skip_proj": 3.0}
    out["boost_keys"] = np.array(list(boost.keys()))
    out["boost_vals"] = np.array(list(boost.values()))
    out.update(model_case("ngpb", m, xn, d * 1.7, g, boost=boost))
    save("g5_ngp.npz", **out)


# ---------------------------------------------------------------- G6 compositing
def g6(g2o):
    g = torch.Generator().manual_seed(6)
    out = {}
    for tag, n in (("a", 64), ("c", 17)):
        z = g2o[f"{tag}_z"][:2, :4].contiguous()
        rgb = torch.rand(2, 4, n, 3, generator=g)
        sigma = torch.rand(2, 4, n, generator=g) * 3
        sigma[0, 0, :] = 0.0
        sigma[0, 1, 5] = 1e-6
        sigma[0, 2, 7:] = 50.0
        sigma[1, 0, -1] = 0.0
        tgt = torch.rand(2, 4, 3, generator=g)
        for wb in (False, True):
            r = rgb.clone().requires_grad_(True)
            s = sigma.clone().requires_grad_(True)
            c, w = Camera.volume_render(r, s, z, white_bkgd=wb)
            loss = torch.nn.functional.mse_loss(c, tgt) + 0.05 * (1 - w.sum(dim=-1)).abs().mean()
            loss.backward()
            k = f"{tag}{int(wb)}"
            out.update({f"{k}_c": c.detach(), f"{k}_w": w.detach(), f"{k}_drgb": r.grad, f"{k}_dsigma": s.grad,
                        f"{k}_loss": loss.detach()})
        out.update({f"{tag}_rgb": rgb, f"{tag}_sigma": sigma, f"{tag}_z": z, f"{tag}_tgt": tgt})
    # legacy free function (1e10 tail, colour only): src/utils/get_rays.py:198-236
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        out["legacy_c"] = legacy.volume_render(out["a_rgb"], out["a_sigma"], out["a_z"])
    save("g6_composite.npz", **out)


# ---------------------------------------------------------------- G7 analytic scene
def g7():
    """Ground-truth renders of the analytic torus+sphere field (the scene of
    src/train_nerf.py:48-66 with an indexable focal), small enough to ship."""
    H = W = 24
    n = 32
    views = []
    for k in range(6):
        ang = 2 * np.pi * k / 6
        eye = torch.tensor([3.0 * np.cos(ang), 3.0 * np.sin(ang), 1.0], dtype=torch.float32)
        # the reference's look-at camera faces away from its target (rays go along -forward),
        # so aim it at the mirrored point to see the object
        cam = Camera(eye=eye, target=2 * eye, focal=(30.0, 30.0), H=H, W=W, up=torch.tensor([0.0, 0.0, 1.0]))
        ro, rd = cam.get_rays()
        pts, z = Camera.sample_points_along_rays(ro, rd, 1.0, 5.0, n, perturb=False)
        rgb, sigma = legacy.evaluate_fake_radiance_field_complex(pts)
        img, _ = Camera.volume_render(rgb, sigma * 5.0, z)
        views.append((cam.c2w, img))
    save("g7_scene.npz", c2w=torch.stack([v[0] for v in views]), img=torch.stack([v[1] for v in views]),
         cfg=np.array([H, W, 30.0, 1.0, 5.0, n]))


# ---------------------------------------------------------------- G8 the older free functions
def g8():
    """src/utils/get_rays.py: compute_stable_up, look_at_camera, get_rays ([W,H,3], +z forward),
    sample_points_along_rays (no perturbation), volume_render (1e10 tail) and the three analytic fields."""
    import contextlib
    import io
    eye, tgt = torch.tensor([2.0, -1.5, 1.0]), torch.tensor([0.1, 0.2, 0.0])
    up = legacy.compute_stable_up(eye, tgt)
    up_vert = legacy.compute_stable_up(torch.tensor([0.0, -3.0, 0.0]), torch.tensor([0.0, 0.0, 0.0]))
    c2w = legacy.look_at_camera(eye, tgt, up)
    H, W, focal = 12, 20, 25.0
    ro, rd = legacy.get_rays(H, W, focal, c2w)
    pts, z = legacy.sample_points_along_rays(ro, rd, 1.0, 5.0, 16)
    out = dict(eye=eye, tgt=tgt, up=up, up_vert=up_vert, c2w=c2w, hwf=np.array([H, W, focal]), rays_o=ro.contiguous(),
               rays_d=rd, pts=pts, z=z.contiguous(), cfg=np.array([1.0, 5.0, 16]))
    for name in ("complex", "sphere", "torus"):
        rgb, sigma = getattr(legacy, "evaluate_fake_radiance_field_" + name)(pts)
        with contextlib.redirect_stdout(io.StringIO()):
            img = legacy.volume_render(rgb.contiguous(), sigma, z)
        out.update({f"{name}_rgb": rgb.contiguous(), f"{name}_sigma": sigma, f"{name}_img": img})
    save("g8_legacy.npz", **out)


# ---------------------------------------------------------------- G9 training trajectory, 100x100 / 64 samples
def g9(steps=320, ckpt=None):
    """The reference's own pieces trained on the analytic scene at the size of its synthetic trainer
    (src/train_nerf.py:91-92: 100 x 100 rays, 64 samples; 40 views on a circle, :53-54): NeRF() from
    torch.manual_seed(0), Camera.sample_points_along_rays(perturb=True) under torch.manual_seed(1000 + step),
    chunked model calls, Camera.volume_render, mse_loss, clip_grad_norm_(1.0) (:165), Adam(lr=5e-4) (:86).
    Recorded: loss of every step, PSNR of a held-out view at a few step counts.  The ground-truth images are
    not stored (the far side re-renders the analytic field); their checksums are.  ``ckpt``: a file to
    checkpoint to every 10 steps and resume from (the run takes an hour; every step reseeds the generator,
    so a resumed run is the same run)."""
    import time
    H = W = 100
    n, near, far, boost = 64, 1.0, 5.0, 5.0
    focal = (100.0, 100.0)
    up = torch.tensor([0.0, 0.0, 1.0])

    def pose(ang, height):
        eye = torch.tensor([2.0 * np.cos(ang), 2.0 * np.sin(ang), height], dtype=torch.float32)
        # look-at cameras face away from their target (rays go along -forward): aim at the mirrored point
        return Camera(eye=eye, target=2 * eye, focal=focal, H=H, W=W, up=up)

    cams = [pose(float(a), 2.0) for a in torch.linspace(0, 2 * torch.pi, steps=40)] + [pose(0.7, 1.5)]
    rays, imgs = [], []
    for cam in cams:
        ro, rd = cam.get_rays()
        pts, z = Camera.sample_points_along_rays(ro, rd, near, far, n, perturb=False)
        rgb, sigma = legacy.evaluate_fake_radiance_field_complex(pts)
        img, _ = Camera.volume_render(rgb, sigma * boost, z)
        rays.append((ro, rd))
        imgs.append(img)

    def run_model(model, pts, dirs, chunk=65536):
        rgbs, sigmas = [], []
        for i in range(0, pts.shape[0], chunk):
            r, s = model(pts[i:i + chunk], dirs[i:i + chunk])
            rgbs.append(r)
            sigmas.append(s)
        return torch.cat(rgbs), torch.cat(sigmas)

    def render(model, ro, rd, perturb):
        pts, z = Camera.sample_points_along_rays(ro, rd, near, far, n, perturb=perturb)
        vd = (rd / rd.norm(dim=-1, keepdim=True))[:, :, None, :].expand(H, W, n, 3)
        rgb, sigma = run_model(model, pts.reshape(-1, 3), vd.reshape(-1, 3))
        return Camera.volume_render(rgb.view(H, W, n, 3), sigma.view(H, W, n), z)[0]

    def psnr(x, y):
        return float(-10.0 * torch.log10(torch.mean((x - y) ** 2)))

    import contextlib
    import io
    torch.manual_seed(0)
    model = NeRF()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    losses, norms, marks, psnrs = [], [], [], []
    first = 0
    if ckpt and os.path.exists(ckpt):
        st = torch.load(ckpt, weights_only=True)
        model.load_state_dict(st["model"])
        opt.load_state_dict(st["opt"])
        losses, norms, marks, psnrs, first = st["losses"], st["norms"], st["marks"], st["psnrs"], st["step"]
        print("resumed at step", first, flush=True)
    t0 = time.time()
    for step in range(first, steps):
        v = step % 40
        torch.manual_seed(1000 + step)
        with contextlib.redirect_stdout(io.StringIO()):   # NeRF.forward prints on NaN
            img = render(model, rays[v][0], rays[v][1], True)
        loss = torch.nn.functional.mse_loss(img, imgs[v])
        opt.zero_grad()
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)))
        opt.step()
        losses.append(float(loss.detach()))
        if (step + 1) in (40, 80, 160, 240, steps):
            with torch.no_grad():
                held = render(model, rays[40][0], rays[40][1], False)
            marks.append(step + 1)
            psnrs.append(psnr(held, imgs[40]))
            print(f"step {step + 1}: loss {losses[-1]:.6f} held-out psnr {psnrs[-1]:.3f} ({time.time() - t0:.0f}s)", flush=True)
        if ckpt and (step + 1) % 10 == 0:
            torch.save({"model": model.state_dict(), "opt": opt.state_dict(), "losses": losses, "norms": norms,
                        "marks": marks, "psnrs": psnrs, "step": step + 1}, ckpt + ".tmp")
            os.replace(ckpt + ".tmp", ckpt)
    save("g9_train.npz", c2w=torch.stack([c.c2w for c in cams]), cfg=np.array([H, W, focal[0], near, far, n, boost]),
         img_sum=np.array([float(i.double().sum()) for i in imgs]), img_sample=torch.stack([i[::10, ::10] for i in imgs]),
         loss=np.array(losses), grad_norm=np.array(norms), psnr_steps=np.array(marks), psnr=np.array(psnrs))


# ---------------------------------------------------------------- G10 training trajectories under LR schedules
def g10():
    """Two short runs of the reference's own pieces with the learning-rate schedules its loops use.
    (a) NeRF() on the analytic scene at 50 x 50 rays, 64 samples, epochs of 8 views: Adam(lr=5e-4),
    clip_grad_norm_(1.0), ExponentialLR(gamma=0.98) stepped once per epoch (src/train_nerf.py:86-87,165,204),
    80 steps.  (b) FastNGPNeRF() on the same field shrunk into the unit cube at 32 x 32 rays, 64 samples, epochs of
    2 views: Adam(lr=1e-2, eps=1e-15), loss = mse + 0.05 mean|1 - sum w| (src/train_nerf_ngp.py:55,221-222),
    CosineAnnealingLR(T_max=30, eta_min=1e-4) stepped once per epoch (:62,375), 40 steps, float32 (the loop's
    autocast is CUDA-only).  Recorded: the loss and the learning rate of every step."""
    import contextlib
    import io
    import time
    up = torch.tensor([0.0, 0.0, 1.0])

    def make_views(count, radius, height, focal, H, W, near, far, n, shrink, boost):
        cams, rays, imgs = [], [], []
        for a in torch.linspace(0, 2 * torch.pi, steps=count + 1)[:count]:
            eye = torch.tensor([radius * np.cos(float(a)), radius * np.sin(float(a)), height], dtype=torch.float32)
            cam = Camera(eye=eye, target=2 * eye, focal=focal, H=H, W=W, up=up)   # faces the origin (see g9)
            ro, rd = cam.get_rays()
            pts, z = Camera.sample_points_along_rays(ro, rd, near, far, n, perturb=False)
            rgb, sigma = legacy.evaluate_fake_radiance_field_complex(pts * shrink)
            img, _ = Camera.volume_render(rgb, sigma * boost, z)
            cams.append(cam); rays.append((ro, rd)); imgs.append(img)
        return cams, rays, imgs

    def render(model, ro, rd, near, far, n, chunk):
        H, W = ro.shape[:2]
        pts, z = Camera.sample_points_along_rays(ro, rd, near, far, n, perturb=True)
        vd = (rd / rd.norm(dim=-1, keepdim=True))[:, :, None, :].expand(H, W, n, 3)
        pf, vf = pts.reshape(-1, 3), vd.reshape(-1, 3)
        rgbs, sigmas = [], []
        for i in range(0, pf.shape[0], chunk):
            r, s = model(pf[i:i + chunk], vf[i:i + chunk])
            rgbs.append(r); sigmas.append(s)
        return Camera.volume_render(torch.cat(rgbs).view(H, W, n, 3), torch.cat(sigmas).view(H, W, n), z)

    out = {}
    t0 = time.time()
    # (a) vanilla + ExponentialLR
    H = W = 50
    n, near, far, boost, views, steps = 64, 1.0, 5.0, 5.0, 8, 80
    cams, rays, imgs = make_views(views, 2.0, 2.0, (50.0, 50.0), H, W, near, far, n, 1.0, boost)
    torch.manual_seed(0)
    model = NeRF()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.98)
    losses, lrs = [], []
    for step in range(steps):
        v = step % views
        torch.manual_seed(2000 + step)
        with contextlib.redirect_stdout(io.StringIO()):
            img, _ = render(model, rays[v][0], rays[v][1], near, far, n, 65536)
        loss = torch.nn.functional.mse_loss(img, imgs[v])
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        losses.append(float(loss.detach()))
        if v == views - 1:
            sched.step()
        if (step + 1) % 8 == 0:
            print(f"g10a step {step + 1}: loss {losses[-1]:.6f} lr {lrs[-1]:.6e} ({time.time() - t0:.0f}s)", flush=True)
    out.update(a_c2w=torch.stack([c.c2w for c in cams]), a_cfg=np.array([H, W, 50.0, near, far, n, boost, views, 1.0]),
               a_img_sum=np.array([float(i.double().sum()) for i in imgs]), a_loss=np.array(losses), a_lr=np.array(lrs))
    # (b) hash-grid model + CosineAnnealingLR
    H = W = 32
    n, near, far, shrink, views, steps = 64, 0.05, 1.6, 2.5, 2, 40
    boost = 5.0 * shrink
    cams, rays, imgs = make_views(views, 0.8, 0.5, (32.0, 32.0), H, W, near, far, n, shrink, boost)
    torch.manual_seed(0)
    model = FastNGPNeRF()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, eps=1e-15)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=30, eta_min=1e-4)
    losses, lrs = [], []
    for step in range(steps):
        v = step % views
        torch.manual_seed(3000 + step)
        img, w = render(model, rays[v][0], rays[v][1], near, far, n, 16384)
        loss = torch.nn.functional.mse_loss(img, imgs[v]) + 0.05 * (1 - w.sum(dim=-1)).abs().mean()
        opt.zero_grad()
        loss.backward()
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        losses.append(float(loss.detach()))
        if v == views - 1:
            sched.step()
        if (step + 1) % 8 == 0:
            print(f"g10b step {step + 1}: loss {losses[-1]:.6f} lr {lrs[-1]:.6e} ({time.time() - t0:.0f}s)", flush=True)
    out.update(b_c2w=torch.stack([c.c2w for c in cams]), b_cfg=np.array([H, W, 32.0, near, far, n, boost, views, shrink]),
               b_img_sum=np.array([float(i.double().sum()) for i in imgs]), b_loss=np.array(losses), b_lr=np.array(lrs))
    save("g10_sched.npz", **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = [a for a in sys.argv[2:]] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8"]   # g9 takes ~40 min: by name only
    o1 = g1() if {"g1", "g2", "g6"} & set(which) else None
    o2 = g2(o1) if {"g2", "g6"} & set(which) else None
    if "g3" in which: g3()
    if "g4" in which: g4()
    if "g5" in which: g5()
    if "g6" in which: g6(o2)
    if "g7" in which: g7()
    if "g8" in which: g8()
    if "g9" in which: g9()
    if "g10" in which: g10()
