"""Dev check (CPU, float64): the collapsed PWAM algebra the fused HIP path implements equals autograd on the plain formulation
(reference lib/backbone.py:1265-1278, 1329-1372, 604-611, 669).

Collapse (per sample; T pixels, C channels, J word slots):
  w = P (V Wo^T) + bo with P = softmax over words [T, J]   ->  IN(w) = (P - Pbar) VW'      with VW' = (V Wo^T) * rstd_w (per channel),
  mean / variance of w over the T pixels from Pbar [J] and Cov(P) [J, J] only -- w itself is never formed.
  IN(q) folded into the keys:  S = q K''^T + s0,  K'' = alpha * rstd_q * K,  s0 = maskbias - mu_q K''^T.
Backward needs, besides GEMMs, only [J, C]- and [J, J]-sized side matrices: H = P^T dwhat, s = colsum(dwhat), G = dS^T q, colsum(dS).
"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
dt = torch.float64
B, T, C, J, nl = 2, 37, 16, 8, 5
eps = 1e-5


def inorm(z):
    mu = z.mean(1, keepdim=True)
    var = z.var(1, unbiased=False, keepdim=True)
    return (z - mu) / torch.sqrt(var + eps)


def rnd(*s):
    return torch.randn(*s, dtype=dt)


x = rnd(B, T, C).requires_grad_(True)
Wv, bv, Wq, bq, Wo, bo, Wm, bm = (rnd(C, C).requires_grad_(True), rnd(C).requires_grad_(True), rnd(C, C).requires_grad_(True), rnd(C).requires_grad_(True),
                                  rnd(C, C).requires_grad_(True), rnd(C).requires_grad_(True), rnd(C, C).requires_grad_(True), rnd(C).requires_grad_(True))
W1, W2 = (0.3 * rnd(C, C)).requires_grad_(True), (0.3 * rnd(C, C)).requires_grad_(True)
Kl, Vl = rnd(B, J, C), rnd(B, J, C)
valid = torch.zeros(B, J, dtype=dt)
valid[0, :nl] = 1
valid[1, :nl - 2] = 1
Kl = (Kl * valid[..., None]).requires_grad_(True)
Vl = (Vl * valid[..., None]).requires_grad_(True)
maskbias = torch.full((B, J), -1e4, dtype=dt)
maskbias[:, :nl] = 1e4 * valid[:, :nl] - 1e4
alpha = C ** -0.5

# ---- plain formulation (what the reference computes) ----
vis = F.gelu(x @ Wv.T + bv)
q = inorm(x @ Wq.T + bq)
S = alpha * q @ Kl.transpose(1, 2) + maskbias[:, None, :]
S = S[..., :nl]
P = torch.softmax(S, -1)
L = P @ Vl[:, :nl]
lang = inorm(L @ Wo.T + bo)
mm = vis * lang
r = F.gelu(mm @ Wm.T + bm)
g2 = F.relu(r @ W1.T) @ W2.T
xo = x + torch.tanh(g2) * r
dr_out, dxo = rnd(B, T, C), rnd(B, T, C)
loss = (r * dr_out).sum() + (xo * dxo).sum()
params = [x, Wv, bv, Wq, bq, Wo, bo, Wm, bm, W1, W2, Kl, Vl]
ref = torch.autograd.grad(loss, params)
ref = dict(zip("x Wv bv Wq bq Wo bo Wm bm W1 W2 K V".split(), ref))

# ---- collapsed forward ----
with torch.no_grad():
    vpre = x @ Wv.T + bv
    qr = x @ Wq.T + bq
    mu_q = qr.mean(1)
    rq = 1.0 / torch.sqrt(qr.var(1, unbiased=False) + eps)                        # [B, C]
    K2 = alpha * rq[:, None, :] * Kl                                              # [B, J, C]
    s0 = maskbias - torch.einsum("bc,bjc->bj", mu_q, K2)
    S2 = torch.einsum("btc,bjc->btj", qr, K2) + s0[:, None, :]
    Pj = torch.zeros(B, T, J, dtype=dt)
    Pj[..., :nl] = torch.softmax(S2[..., :nl], -1)                                # word slots >= n_l: probability exactly 0
    Pbar = Pj.mean(1)                                                             # [B, J]
    PP = torch.einsum("btj,btk->bjk", Pj, Pj)
    Cov = PP / T - Pbar[:, :, None] * Pbar[:, None, :]
    VW = Vl @ Wo.T                                                                # [B, J, C]
    var_w = torch.einsum("bjc,bjk,bkc->bc", VW, Cov, VW)
    rw = 1.0 / torch.sqrt(var_w + eps)
    VWs = VW * rw[:, None, :]
    what = (Pj - Pbar[:, None, :]) @ VWs
    assert torch.allclose(what, lang, atol=1e-9), (what - lang).abs().max()
    visc = F.gelu(vpre)
    mmc = visc * what
    rpre = mmc @ Wm.T + bm
    rc = F.gelu(rpre)
    g1 = F.relu(rc @ W1.T)
    g2c = g1 @ W2.T
    assert torch.allclose(x + torch.tanh(g2c) * rc, xo, atol=1e-9)

    # ---- collapsed backward ----
    def gelu_grad(z):
        return 0.5 * (1 + torch.erf(z / 2 ** 0.5)) + z * torch.exp(-0.5 * z * z) / (2 * torch.pi) ** 0.5

    th = torch.tanh(g2c)
    dg2 = dxo * rc * (1 - th * th)
    dr = dr_out + dxo * th
    dW2 = torch.einsum("btn,btk->nk", dg2, g1)
    dpre1 = (dg2 @ W2) * (g1 > 0)
    dW1 = torch.einsum("btn,btk->nk", dpre1, rc)
    dr = dr + dpre1 @ W1
    drpre = dr * gelu_grad(rpre)
    dWm = torch.einsum("btn,btk->nk", drpre, mmc)
    dbm = drpre.sum((0, 1))
    dmm = drpre @ Wm
    dvpre = dmm * what * gelu_grad(vpre)
    dwh = dmm * visc
    # side matrices (TN GEMMs over the pixels)
    H = torch.einsum("btj,btc->bjc", Pj, dwh)
    s = dwh.sum(1)
    a = s / T
    b = torch.einsum("bjc,bjc->bc", VWs, H - Pbar[:, :, None] * s[:, None, :]) / T
    dVW = rw[:, None, :] * (H - Pj.sum(1)[:, :, None] * a[:, None, :] - T * b[:, None, :] * torch.einsum("bjk,bkc->bjc", Cov, VWs))
    dV = dVW @ Wo
    dWo = torch.einsum("bjc,bje->ce", dVW, Vl)
    u = torch.einsum("bjc,bc->bj", VWs, a)
    Q = torch.einsum("bkc,bc,bjc->bkj", VWs, b, VWs)
    cst = torch.einsum("bk,bkj->bj", Pbar, Q) - u
    dP = torch.einsum("btc,bjc->btj", dwh, VWs) - torch.einsum("btk,bkj->btj", Pj, Q) + cst[:, None, :]
    dS = Pj * (dP - (dP * Pj).sum(-1, keepdim=True))
    G = torch.einsum("btj,btc->bjc", dS, qr)                                       # raw q
    sdS = dS.sum(1)                                                                # [B, J]
    Ghat = (G - sdS[:, :, None] * mu_q[:, None, :]) * rq[:, None, :]               # = dS^T qhat
    dK = alpha * Ghat
    a2 = alpha * torch.einsum("bj,bjc->bc", sdS, Kl) / T
    b2 = alpha * torch.einsum("bjc,bjc->bc", Kl, Ghat) / T
    c1 = rq * rq * b2                                                              # dq = dS Kq' + c0 - q * c1
    c0 = -rq * a2 + mu_q * c1
    dq = dS @ K2 + c0[:, None, :] - qr * c1[:, None, :]
    dx = dxo + dvpre @ Wv + dq @ Wq
    got = dict(x=dx, Wv=torch.einsum("btn,btk->nk", dvpre, x), bv=dvpre.sum((0, 1)), Wq=torch.einsum("btn,btk->nk", dq, x), bq=dq.sum((0, 1)),
               Wo=dWo, bo=torch.zeros(C, dtype=dt), Wm=dWm, bm=dbm, W1=dW1, W2=dW2, K=dK, V=dV)
    for k in ref:
        e = float((got[k] - ref[k]).abs().max())
        sc = float(ref[k].abs().max()) + 1e-12
        print(f"{k:3s} max|err| {e:.3e} (scale {sc:.3e})")
        assert e < 1e-8 * max(sc, 1.0), k
print("collapsed PWAM algebra: forward and every gradient match autograd")
