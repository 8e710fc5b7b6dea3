"""Observation-side depth -> point-cloud step, mirroring the reference's
`utils/depth2tsdf.py:TSDFVolume` for the part the point-cloud learner consumes:

    tsdf = TSDFVolume(device, size, resolution, _vol_origin)
    tsdf.register_camera(cam_pose (m,4,4), cam_intr (3,3), im_h, im_w, num_env)     # depth2tsdf.py:30-66
    clouds = tsdf.depth2pc(depth_im (b, m, h, w))  -> (b, 1024, 3)                  # depth2tsdf.py:136-173

`depth2pc` = back-projection + workspace crop (pm_depth_backproject_f32), compaction of the zeroed points
(pm_depth_compact_f32) and farthest point sampling with pytorch3d's defaults (pm_fps_varlen_f32: start index 0,
lowest-index ties -- the same points as sampling the full cloud; the reference calls
`pytorch3d.ops.sample_farthest_points(world_cld, K=1024)`, depth2tsdf.py:160), both on the GPU.
`integrate` (depth -> TSDF volume for the Conv3D students, depth2tsdf.py:68-86) is pm_tsdf_integrate_f32 over voxel ->
pixel tables built at registration time; `sparse_voxel` (depth2tsdf.py:88-120, the 'depth_sparse' observation) selects the
surface band, samples it with pm_fps_varlen_f32 and gathers (x, y, z, tsdf).  The marching-cubes `extract_point_cloud`
is outside this build's scope: it raises NotImplementedError.
"""
import numpy as np
import torch

from . import ops


class TSDFVolume(object):
    def __init__(self, device, size=0.5, resolution=50, _vol_origin=(-0.25, -0.25, -0.0503)):
        self._size = size
        self._resolution = resolution
        self._voxel_size = self._size / self._resolution
        self._sdf_trunc = 4 * self._voxel_size
        self.device = device
        self._vol_origin = torch.tensor(list(_vol_origin), dtype=torch.float32, device=device)
        self.default_tsdf = 1
        self._ws = ops.Workspace(device)

    def register_camera(self, cam_pose, cam_intr, im_h, im_w, num_env):
        """cam_pose (m,4,4) camera->world, cam_intr (3,3), image size, env count (depth2tsdf.py:30-66).
        Registration-time (not per step): the voxel -> pixel tables of `integrate`, with the reference's own
        tensor expressions on the host (depth2tsdf.py:41-60)."""
        cam_pose = torch.as_tensor(np.asarray(cam_pose), dtype=torch.float32)
        self.registered_shape = (num_env, cam_pose.shape[0], im_h, im_w)
        self.cam_pose = cam_pose.to(self.device).contiguous()                      # (m,4,4); not repeated per env
        self.cam_intr = cam_intr
        res = self._resolution
        ax = torch.arange(0, res)
        xv, yv, zv = torch.meshgrid(ax, ax, ax, indexing="ij")
        vox = torch.stack([xv.flatten(), yv.flatten(), zv.flatten()], dim=1).long()
        world_c = self._vol_origin.cpu() + (self._voxel_size * vox)
        world_c = world_c[None, ...].repeat(cam_pose.shape[0], 1, 1)
        cam_c = torch.bmm(world_c - cam_pose[:, :3, 3].unsqueeze(-2), cam_pose[:, :3, :3])
        fx, fy = float(cam_intr[0][0]), float(cam_intr[1][1])
        cx, cy = float(cam_intr[0][2]), float(cam_intr[1][2])
        pix_z = cam_c[..., 2]
        pix_x = torch.round((cam_c[..., 0] * fx / cam_c[..., 2]) + cx).long()
        pix_y = torch.round((cam_c[..., 1] * fy / cam_c[..., 2]) + cy).long()
        valid = (pix_x >= 0) & (pix_x < im_w) & (pix_y >= 0) & (pix_y < im_h) & (pix_z > 0)
        idx = torch.where(valid, pix_y * im_w + pix_x, torch.full_like(pix_x, -1))
        self.pix_z = pix_z.contiguous().to(self.device)
        self.valid_pix = valid.to(self.device)
        self._pix_idx = idx.to(torch.int32).contiguous().to(self.device)

    def depth2pc(self, depth_im, K=1024):
        """depth_im (b, m, h, w) float32 on the device -> (b, K, 3) world-frame clouds."""
        assert tuple(depth_im.shape) == tuple(self.registered_shape)
        intr = self.cam_intr
        # the reference's naming is swapped (cam_cx pairs with the column map, depth2tsdf.py:146-149); kept as is
        cam_cx, cam_cy = float(intr[0][2]), float(intr[1][2])
        cam_fx, cam_fy = float(intr[0][0]), float(intr[1][1])
        lo = self._vol_origin.cpu().numpy().astype(np.float32)
        hi = (np.float32(self._size) + lo).astype(np.float32)                       # `self._size + self._vol_origin`
        world = ops.depth_backproject(depth_im.float().contiguous(), self.cam_pose, cam_fx, cam_fy, cam_cx, cam_cy, lo, hi)
        # the zeroed out-of-crop points are ONE candidate for the sampler: compact them away first (same selected
        # points, 3-20x less to read per round), then sample the variable-length clouds
        compact, lengths = ops.depth_compact(world)
        idx = ops.fps_varlen(compact, lengths, K, self._ws)                        # (b, K) int32 into `compact`
        return ops.group_points(compact, idx.view(idx.shape[0], K, 1)).view(idx.shape[0], K, 3)

    def integrate(self, depth_im):
        """depth_im (b, m, h, w) -> TSDF volume (b, res, res, res) (depth2tsdf.py:68-86)."""
        assert tuple(depth_im.shape) == tuple(self.registered_shape)
        vol = ops.tsdf_integrate(depth_im.float().contiguous(), self._pix_idx, self.pix_z, self._sdf_trunc, self.default_tsdf)
        self._tsdf_vol = vol.view(depth_im.shape[0], self._resolution, self._resolution, self._resolution)
        return self._tsdf_vol

    def sparse_voxel(self, depth_im, K=1024):
        """depth_im (b, m, h, w) -> (b, K, 4) rows (x, y, z, tsdf): the 'depth_sparse' observation (depth2tsdf.py:88-120):
        integrate, keep the voxels with -0.2 < tsdf < 0.2, farthest-point-sample their integer coordinates."""
        vol = self.integrate(depth_im)
        return ops.tsdf_sparse_voxel(vol.contiguous(), K, -0.2, 0.2, self._ws)

    def extract_point_cloud(self):
        raise NotImplementedError("marching cubes is outside this build's scope (SURVEY.md §8f)")
