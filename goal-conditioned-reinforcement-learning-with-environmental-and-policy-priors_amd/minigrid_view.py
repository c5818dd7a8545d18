"""Torch front end of mg_gen_obs (include/minigrid_view.h): the agent view of any MiniGridEnv-style world kept
as structure-of-arrays planes on the GPU -- MiniGridEnv.gen_obs / gen_obs_grid of the reference
(gym_minigrid/minigrid.py:1443-1496) with occlusion (process_vis :795-832), any agent direction, any odd view
size and the carried object, for N envs per launch.  No CPU fallback."""
import ctypes as C

import torch

from . import _lib

MG_MAX_VIEW = 31


def planes_from_encoded(enc):
    """Grid.encode() arrays [N][W][H][3] (x-major, as the reference returns them) -> (type, colour, state)
    planes uint8[N][H*W] with cell (x, y) at y*W + x, the layout of Grid.grid (minigrid.py:562-569)."""
    e = torch.as_tensor(enc)
    assert e.dim() == 4 and e.shape[-1] == 3
    p = e.permute(0, 2, 1, 3).contiguous()                   # [N][H][W][3]
    N, H, W, _ = p.shape
    return tuple(p[..., k].reshape(N, H * W).contiguous() for k in range(3))


def _p(t, dtype):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, "expected contiguous %s device tensor" % dtype
    return C.c_void_p(t.data_ptr())


def gen_obs(type_plane, colour_plane, state_plane, width, height, agent_x, agent_y, agent_dir, view_size,
            see_through_walls=False, carrying=None, want_mask=True, out=None):
    """-> (image uint8[N,V,V,3], vis_mask uint8[N,V,V] or None).  Planes uint8[N, H*W]; agent_* int32[N];
    carrying uint8[N,3] (type 0 = nothing) or None."""
    N = type_plane.shape[0]
    V = int(view_size)
    assert 1 <= V <= MG_MAX_VIEW
    assert type_plane.shape == (N, width * height) and colour_plane.shape == (N, width * height)
    dev = type_plane.device
    image = out if out is not None else torch.empty((N, V, V, 3), dtype=torch.uint8, device=dev)
    assert image.shape == (N, V, V, 3)
    mask = torch.empty((N, V, V), dtype=torch.uint8, device=dev) if want_mask else None
    _lib.check(_lib.lib().mg_gen_obs(
        _p(type_plane, torch.uint8), _p(colour_plane, torch.uint8), _p(state_plane, torch.uint8), N, int(width),
        int(height), _p(agent_x, torch.int32), _p(agent_y, torch.int32), _p(agent_dir, torch.int32),
        _p(carrying, torch.uint8), V, int(bool(see_through_walls)), _p(image, torch.uint8), 0, _p(mask, torch.uint8),
        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mg_gen_obs")
    return image, mask


def step(type_plane, state_plane, width, height, action, agent_x, agent_y, agent_dir, step_count, max_steps):
    """MiniGridEnv.step of the base class for N envs (mg_step): agent_x / agent_y / step_count (int32[N]) are
    updated in place; -> (reward float64[N], terminated uint8[N], truncated uint8[N], error int32[N])."""
    N = type_plane.shape[0]
    dev = type_plane.device
    reward = torch.empty(N, dtype=torch.float64, device=dev)
    term = torch.empty(N, dtype=torch.uint8, device=dev)
    trunc = torch.empty(N, dtype=torch.uint8, device=dev)
    err = torch.empty(N, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().mg_step(
        _p(type_plane, torch.uint8), _p(state_plane, torch.uint8), N, int(width), int(height), _p(action, torch.int32),
        _p(agent_x, torch.int32), _p(agent_y, torch.int32), _p(agent_dir, torch.int32), _p(step_count, torch.int32),
        int(max_steps), _p(reward, torch.float64), _p(term, torch.uint8), _p(trunc, torch.uint8), _p(err, torch.int32),
        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mg_step")
    return reward, term, trunc, err
