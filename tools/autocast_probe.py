import torch
x = (torch.randn(4, 64, device="cuda") * 1.5).bfloat16()
with torch.autocast("cuda", dtype=torch.bfloat16):
    a = torch.abs(x); print("abs", a.dtype)
    m = torch.max(a, dim=-1, keepdim=True)[0].expand_as(x); print("max", m.dtype)
    t1 = m + 1e-6; print("max+1e-6", t1.dtype)
    s = 127 / t1; print("int/tensor", s.dtype)
    r = t1.reciprocal(); print("reciprocal", r.dtype)
    p = x * s; print("x*s", p.dtype)
    q = torch.round(p); print("round", q.dtype)
    t2 = s + 1e-6; print("s+1e-6", t2.dtype)
    y = q.div(t2); print("div", y.dtype)
    # does it equal a pure-fp32 recipe with the bf16-rounded t1?
    t1f = t1.float()
    s_ref = (1.0 / t1f) * 127
    y_ref = torch.round(x.float() * s_ref) / (s_ref + 1e-6)
    print("matches fp32 recipe:", torch.equal(y, y_ref), "s equal:", torch.equal(s, s_ref))
    s_ref2 = (t1f.reciprocal()) * 127.0
    print("s equal recip*127:", torch.equal(s, s_ref2))
    # asym pieces
    n = (x - x.min(dim=-1, keepdim=True)[0]) / ((x.max(dim=-1, keepdim=True)[0] - x.min(dim=-1, keepdim=True)[0]) + 1e-8); print("asym n", n.dtype)
    print("mean", x.abs().mean(dim=1).dtype, "sign", torch.sign(x).dtype, "clamp", torch.clamp(x, -1, 1).dtype)
    g = torch.randn(4, 64, device="cuda")
    gi = g.clone(); gi[x.ge(2.0)] = 0; print("ste grad dtype", gi.dtype)
