"""Host-side index tables that turn Swin's data-movement ops into GEMM row gathers / scatters.

The reference moves data with pad -> roll -> window_partition ... window_reverse -> roll -> crop
(lib/backbone.py:204-237, 33-62), rebuilds the SW-MSA mask with Python loops on every stage call
(:634-652) and concatenates 2x2 neighbours for PatchMerging (:278-283).  Here each of those is a
small int32/int8 table, computed once per (shape, window, shift) on the host with integer
arithmetic, cached on the device, and consumed by the HIP kernels (a_rowmap / c_rowmap / gather /
region arguments of include/lavt_hip.h).  Pure NumPy: testable without a GPU.
"""
import functools

import numpy as np
import torch


def padded(n: int, ws: int) -> int:
    return -(-n // ws) * ws


def window_map_np(B: int, H: int, W: int, ws: int, shift: int) -> np.ndarray:
    """map[m] = source token (b*H*W + y*W + x) of windowed row m, or -1 for zero padding.

    Windowed rows are ordered (b, window-row a, window-col c, p, q) (lib/backbone.py:43-44); the window
    grid lives on the padded, cyclically shifted map: shifted[y] = padded[(y + shift) % Hp] (:213).
    The same table is the scatter map of the reverse path (:228-237)."""
    Hp, Wp = padded(H, ws), padded(W, ws)
    a, p = np.divmod(np.arange(Hp), ws)           # shifted-grid row -> (window row, row in window)
    ys = (np.arange(Hp) + shift) % Hp             # original padded row of each shifted row
    xs = (np.arange(Wp) + shift) % Wp
    src = np.where((ys[:, None] < H) & (xs[None, :] < W), ys[:, None] * W + xs[None, :], -1)       # (Hp, Wp)
    win = src.reshape(Hp // ws, ws, Wp // ws, ws).transpose(0, 2, 1, 3).reshape(-1)                # (nW*N,)
    out = np.where(win[None, :] >= 0, win[None, :] + (np.arange(B) * H * W)[:, None], -1)
    return out.reshape(-1).astype(np.int32)


def window_inverse_np(B: int, H: int, W: int, ws: int, shift: int) -> np.ndarray:
    """inverse of window_map_np on the real tokens: inv[token] = its windowed row (every token sits in exactly one window).  Lets a contraction over
    the windowed rows run over the tokens instead: the zero rows of padded window positions drop out of the reduction."""
    wm = window_map_np(B, H, W, ws, shift)
    inv = np.full(B * H * W, -1, np.int32)
    rows = np.nonzero(wm >= 0)[0]
    inv[wm[rows]] = rows
    assert (inv >= 0).all()
    return inv


def window_pad_rows_np(B: int, H: int, W: int, ws: int, shift: int) -> np.ndarray:
    """the windowed rows that hold padding (window_map_np == -1), ascending"""
    return np.nonzero(window_map_np(B, H, W, ws, shift) < 0)[0].astype(np.int32)


def region_ids_np(H: int, W: int, ws: int, shift: int) -> np.ndarray:
    """int8 [nW][N]: 3*g(row)+g(col) on the padded grid, g = 0 | 1 | 2 for [0,Hp-ws) | [Hp-ws,Hp-shift) | rest
    (lib/backbone.py:636-647).  Two tokens of a window attend to each other iff their ids are equal."""
    Hp, Wp = padded(H, ws), padded(W, ws)

    def g(n):
        r = np.arange(n)
        return np.where(r < n - ws, 0, np.where(r < n - shift, 1, 2))
    ids = 3 * g(Hp)[:, None] + g(Wp)[None, :]
    return ids.reshape(Hp // ws, ws, Wp // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws).astype(np.int8)


def merge_map_np(B: int, H: int, W: int) -> np.ndarray:
    """int32 [B*H2*W2][4]: source tokens of the 2x2 neighbourhood in the order (0,0),(1,0),(0,1),(1,1)
    (lib/backbone.py:278-282); -1 where an odd H/W was zero padded (:274-276)."""
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    i, j = np.meshgrid(np.arange(H2), np.arange(W2), indexing="ij")
    out = np.empty((B, H2, W2, 4), np.int64)
    for q, (dy, dx) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        y, x = 2 * i + dy, 2 * j + dx
        ok = (y < H) & (x < W)
        t = np.where(ok, y * W + x, -1)
        out[..., q] = np.where(t[None] >= 0, t[None] + (np.arange(B) * H * W)[:, None, None], -1)
    return out.reshape(-1, 4).astype(np.int32)


def kv_pad_map_np(B: int, n_l: int, ld: int) -> np.ndarray:
    """row b*n_l + j of the (B*n_l)-row language projections -> row b*ld + j of the zero-padded [B][ld] K/V buffers."""
    return (np.arange(B)[:, None] * ld + np.arange(n_l)[None, :]).reshape(-1).astype(np.int32)


# ---- Video-Swin: (D, H, W) token volumes, (wd, wh, ww) windows ----------------------------------------------------------
def clip_window(size, window, shift=None):
    """lib/video_swin_transformer.py:70-83: on an axis whose extent is <= the window, the window is clipped to it and the shift zeroed."""
    win = tuple(s if s <= w else w for s, w in zip(size, window))
    if shift is None:
        return win
    return win, tuple(0 if s <= w else sh for s, w, sh in zip(size, window, shift))


def window_map3d_np(B: int, D: int, H: int, W: int, win, shift) -> np.ndarray:
    """3-D twin of window_map_np: windowed rows ordered (b, window d/h/w index, position d/h/w in window)
    (lib/video_swin_transformer.py:39-52); source token = ((b*D + z)*H + y)*W + x, -1 = zero padding (:236-241).
    `win` / `shift` are the already clipped values."""
    (wd, wh, ww), (sd, sh, sw) = win, shift
    Dp, Hp, Wp = padded(D, wd), padded(H, wh), padded(W, ww)
    zs, ys, xs = (np.arange(Dp) + sd) % Dp, (np.arange(Hp) + sh) % Hp, (np.arange(Wp) + sw) % Wp
    ok = (zs[:, None, None] < D) & (ys[None, :, None] < H) & (xs[None, None, :] < W)
    src = np.where(ok, (zs[:, None, None] * H + ys[None, :, None]) * W + xs[None, None, :], -1)            # (Dp, Hp, Wp)
    win_rows = src.reshape(Dp // wd, wd, Hp // wh, wh, Wp // ww, ww).transpose(0, 2, 4, 1, 3, 5).reshape(-1)
    out = np.where(win_rows[None, :] >= 0, win_rows[None, :] + (np.arange(B) * D * H * W)[:, None], -1)
    return out.reshape(-1).astype(np.int32)


def region_ids3d_np(D: int, H: int, W: int, win, shift) -> np.ndarray:
    """int8 [nW][N]: 9*g(d) + 3*g(h) + g(w) on the padded volume (compute_mask, lib/video_swin_transformer.py:315-328:
    slices [0,-w) | [-w,-s) | [-s,end) per axis; with s == 0 the last slice is the whole axis)."""
    (wd, wh, ww), (sd, sh, sw) = win, shift
    Dp, Hp, Wp = padded(D, wd), padded(H, wh), padded(W, ww)

    def g(n, w, s):
        r = np.arange(n)
        return np.where(r < n - w, 0, np.where(r < n - s, 1, 2)) if s > 0 else np.full(n, 2)
    ids = 9 * g(Dp, wd, sd)[:, None, None] + 3 * g(Hp, wh, sh)[None, :, None] + g(Wp, ww, sw)[None, None, :]
    return ids.reshape(Dp // wd, wd, Hp // wh, wh, Wp // ww, ww).transpose(0, 2, 4, 1, 3, 5).reshape(-1, wd * wh * ww).astype(np.int8)


@functools.lru_cache(maxsize=256)
def _cached(kind, args, device):
    fn = {"window": window_map_np, "winv": window_inverse_np, "wpad": window_pad_rows_np, "region": region_ids_np, "merge": merge_map_np, "kvpad": kv_pad_map_np,
          "window3d": window_map3d_np, "region3d": region_ids3d_np}[kind]
    return torch.from_numpy(fn(*args)).to(device)


def window_map(B, H, W, ws, shift, device):
    return _cached("window", (B, H, W, ws, shift), str(device))


def window_inverse(B, H, W, ws, shift, device):
    return _cached("winv", (B, H, W, ws, shift), str(device))


def window_pad_rows(B, H, W, ws, shift, device):
    return _cached("wpad", (B, H, W, ws, shift), str(device))


def region_ids(H, W, ws, shift, device):
    return _cached("region", (H, W, ws, shift), str(device))


def merge_map(B, H, W, device):
    return _cached("merge", (B, H, W), str(device))


def kv_pad_map(B, n_l, ld, device):
    return _cached("kvpad", (B, n_l, ld), str(device))


def window_map3d(B, D, H, W, win, shift, device):
    return _cached("window3d", (B, D, H, W, tuple(win), tuple(shift)), str(device))


def region_ids3d(D, H, W, win, shift, device):
    return _cached("region3d", (D, H, W, tuple(win), tuple(shift)), str(device))
