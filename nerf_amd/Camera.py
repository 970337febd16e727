"""Drop-in for the reference's ``Camera`` module (src/Camera.py): same names,
arguments and return shapes, with ray generation, sampling and compositing
running as gfx950 HIP kernels.

Differences a caller can observe: tensors must live on the GPU (there is no CPU
path), outputs are freshly allocated dense tensors where the reference returns
expanded views (``rays_o``, unperturbed ``z_vals``), and
``sample_points_along_rays`` accepts an optional ``t_rand`` so that the uniform
draws can be injected (the reference always calls ``torch.rand_like``).
"""
import torch

from . import ops


def look_at_camera(eye, target, up=None):
    """c2w [3,4] = [right | true_up | forward | eye]  (src/Camera.py:6-30).
    Host-side 3-vector algebra; kept in PyTorch.  Note that get_rays() shoots
    along -forward, exactly as the reference does."""
    eye = eye.float()
    target = target.float()
    forward = target - eye
    forward = forward / torch.norm(forward)
    if up is None:
        up = torch.tensor([0.0, 1.0, 0.0], dtype=eye.dtype, device=eye.device)
    up = up.float()
    if torch.abs(torch.dot(forward, up)) > 0.999:
        up = torch.tensor([0.0, 0.0, 1.0], dtype=eye.dtype, device=eye.device)
    right = torch.cross(up, forward, dim=0)
    right = right / torch.norm(right)
    true_up = torch.cross(forward, right, dim=0)
    return torch.cat([torch.stack([right, true_up, forward], dim=1), eye.view(3, 1)], dim=1)


class Camera:
    """Pinhole camera, constructor as src/Camera.py:35-47 (``focal`` is (fx, fy))."""

    def __init__(self, eye, target, focal, H, W, c2w=None, pp=None, up=None):
        self.eye = eye
        self.target = target
        self.focal = focal
        self.H = H
        self.W = W
        self.principal_point = pp
        self.up = up
        self.c2w = c2w if c2w is not None else look_at_camera(self.eye, self.target, self.up)

    def _intrinsics(self):
        if self.principal_point is not None:
            cx, cy = self.principal_point
        else:
            cx, cy = self.W * 0.5, self.H * 0.5
        return float(self.focal[0]), float(self.focal[1]), float(cx), float(cy)

    def get_rays(self):
        """rays_o, rays_d [H,W,3]  (src/Camera.py:49-72)."""
        fx, fy, cx, cy = self._intrinsics()
        o, d = ops.raygen(self.c2w, fx, fy, cx, cy, int(self.H), int(self.W))
        return o.view(self.H, self.W, 3), d.view(self.H, self.W, 3)

    def get_ray_batch(self, pix):
        """Rays of the int pixel ids ``pix`` (row*W+col), [R,3] each.  Not in the
        reference, which always renders whole images."""
        fx, fy, cx, cy = self._intrinsics()
        return ops.raygen(self.c2w, fx, fy, cx, cy, int(self.H), int(self.W), pix)

    @staticmethod
    def axis_angle_to_matrix(axis, theta):
        """Rodrigues rotation (src/Camera.py:74-96); host-side pose utility."""
        axis = axis / axis.norm(dim=-1, keepdim=True).clamp(min=1e-9)
        ax, ay, az = axis.unbind(-1)
        c, s = torch.cos(theta), torch.sin(theta)
        C = 1.0 - c
        rows = [c + ax * ax * C, ax * ay * C - az * s, ax * az * C + ay * s,
                ay * ax * C + az * s, c + ay * ay * C, ay * az * C - ax * s,
                az * ax * C - ay * s, az * ay * C + ax * s, c + az * az * C]
        return torch.stack(rows, dim=-1).reshape(axis.shape[:-1] + (3, 3))

    @staticmethod
    def look_at_to_matrix(eye, target, up=None):
        """Batched look-at pose [...,3,4] (src/Camera.py:98-125); host-side."""
        if up is None:
            up = torch.tensor([0., 1., 0.], dtype=eye.dtype, device=eye.device)
        forward = target - eye
        forward = forward / forward.norm(dim=-1, keepdim=True).clamp(min=1e-9)
        right = torch.cross(up.expand_as(forward), forward, dim=-1)
        right = right / right.norm(dim=-1, keepdim=True).clamp(min=1e-9)
        true_up = torch.cross(forward, right, dim=-1)
        return torch.cat([torch.stack([right, true_up, forward], dim=-1), eye.unsqueeze(-1)], dim=-1)

    @staticmethod
    def sample_points_along_rays(rays_o, rays_d, near, far, N_samples, perturb=True, t_rand=None):
        """pts [H,W,N,3], z_vals [H,W,N]  (src/Camera.py:127-141).  Like the
        reference, exactly two leading dimensions are expected."""
        H, W = rays_o.shape[:2]
        dev = rays_o.device
        z_lin = torch.linspace(near, far, N_samples).to(dev)   # the reference's CPU linspace values, bit for bit
        if perturb and t_rand is None:
            t_rand = torch.rand(H, W, N_samples, device=dev)
        u = t_rand.reshape(H * W, N_samples) if perturb else None
        z, pts = ops.stratified_sample(rays_o.reshape(H * W, 3), rays_d.reshape(H * W, 3), z_lin, u)
        return pts.view(H, W, N_samples, 3), z.view(H, W, N_samples)

    @staticmethod
    def volume_render(rgb, sigma, z_vals, white_bkgd=False):
        """final_rgb [H,W,3], weights [H,W,N]  (src/Camera.py:143-183); differentiable
        in rgb and sigma through both outputs."""
        H, W, N_samples = sigma.shape
        device = rgb.device
        out, w, _ = ops.composite(rgb, sigma.to(device), z_vals.to(device), white_bkgd)
        return out, w

    def visualize_corner_rays(self, rays_o, rays_d, obj_center, xlim=1, ylim=1, zlim=1):
        """Matplotlib sketch of the four corner rays (src/Camera.py:185-200)."""
        import matplotlib.pyplot as plt
        fig = plt.figure()
        ax = fig.add_subplot(111, projection='3d')
        c = [float(v) for v in obj_center]
        ax.scatter(c[0], c[1], c[2], color='black', label='Origin', s=50)
        for (r, col), colour in zip(((0, 0), (0, -1), (-1, 0), (-1, -1)), 'rgby'):
            ax.quiver(*rays_o[r, col].tolist(), *rays_d[r, col].tolist(), color=colour, length=1.0)
        ax.set_xlim([-xlim, xlim]); ax.set_ylim([-ylim, ylim]); ax.set_zlim([-zlim, zlim])
        ax.set_xlabel('X'); ax.set_ylabel('Y'); ax.set_zlabel('Z')
        ax.set_title("Ray directions from camera corners")
        plt.show()
