"""Plain PyTorch restatement of MGN-spec v1 (TEST INFRASTRUCTURE): an implementation independent of the NumPy oracle --
torch.nn.functional ops, index_add_ for the scatter, autograd for the gradients -- used to cross-check the oracle's forward
and its hand-written reverse mode, and as the fp32 reference of the HIP kernels on the GPU box.
Parameters: the packed vector of include/mgn_hip.h (one leaf tensor, sliced into views), so gradients come back packed."""
import torch
import torch.nn.functional as F


def _take(p, off, shape):
    n = 1
    for s in shape:
        n *= s
    return p[off:off + n].view(*shape), off + n


def _mlp(p, off, x, n_in, L, n_out, ln):
    dims = [n_in, L, L, n_out]
    for i in range(3):
        W, off = _take(p, off, (dims[i], dims[i + 1]))          # row-major [in][out]
        b, off = _take(p, off, (dims[i + 1],))
        x = x @ W + b
        if i < 2:
            x = F.relu(x)
    if ln:
        g, off = _take(p, off, (n_out,))
        be, off = _take(p, off, (n_out,))
        x = F.layer_norm(x, (n_out,), g, be, eps=1e-5)
    return x, off


def forward(p, cfg, nf, ef, senders, receivers, set2=None):
    """p: packed parameters (torch tensor); returns out [N][O]."""
    Fn, Fe, O, L, mps = cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], cfg["mps"]
    s = torch.as_tensor(senders, dtype=torch.long, device=p.device)
    r = torch.as_tensor(receivers, dtype=torch.long, device=p.device)
    N = nf.shape[0]
    off = 0
    v, off = _mlp(p, off, nf, Fn, L, L, True)
    e, off = _mlp(p, off, ef, Fe, L, L, True)
    if set2 is not None:
        ef2, s2, r2 = set2
        s2 = torch.as_tensor(s2, dtype=torch.long, device=p.device)
        r2 = torch.as_tensor(r2, dtype=torch.long, device=p.device)
        e2, off = _mlp(p, off, ef2, cfg["Fe2"], L, L, True)
    for _ in range(mps):
        en, off = _mlp(p, off, torch.cat([v[s], v[r], e], 1), 3 * L, L, L, True)
        aggs = [torch.zeros(N, L, dtype=p.dtype, device=p.device).index_add_(0, r, en)]
        if set2 is not None:
            e2n, off = _mlp(p, off, torch.cat([v[s2], v[r2], e2], 1), 3 * L, L, L, True)
            aggs.append(torch.zeros(N, L, dtype=p.dtype, device=p.device).index_add_(0, r2, e2n))
        vn, off = _mlp(p, off, torch.cat([v] + aggs, 1), (1 + len(aggs)) * L, L, L, True)
        v, e = v + vn, e + en
        if set2 is not None:
            e2 = e2 + e2n
    out, off = _mlp(p, off, v, L, L, O, False)
    assert off == p.numel(), (off, p.numel())
    return out


def step(p_np, cfg, nf, ef, senders, receivers, target, mask, dtype=torch.float64, device="cpu"):
    """(gs, loss) of step! by autograd: loss = mean(sum_o (target - out)^2 over the masked nodes)."""
    p = torch.tensor(p_np, dtype=dtype, device=device, requires_grad=True)
    out = forward(p, cfg, torch.as_tensor(nf, dtype=dtype, device=device), torch.as_tensor(ef, dtype=dtype, device=device), senders, receivers)
    m = torch.as_tensor(mask, dtype=torch.long, device=device)
    err = ((torch.as_tensor(target, dtype=dtype, device=device) - out) ** 2).sum(1)
    loss = err[m].mean()
    loss.backward()
    return p.grad.detach().cpu().numpy(), float(loss.detach())
