"""Probe (GPU box): which fp32 formulation reproduces torch-ROCm's bf16 elementwise ops bit-exactly."""
import numpy as np, torch
import sys
sys.path.insert(0, ".")
from oracle.rl_math import bf16_round
f = np.float32
rs = np.random.RandomState(0)
n = 400000
def bt(x): return torch.from_numpy(x).bfloat16().cuda()
def fma(a, b, c): return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f)
v = bf16_round(np.abs(rs.standard_normal(n)).astype(f) * 1e-3); g = bf16_round(rs.standard_normal(n).astype(f) * 3e-2)
t = bt(v).clone(); t.addcmul_(bt(g), bt(g), value=1 - 0.999); ref = t.float().cpu().numpy()
val = f(1 - 0.999)
for k, c in {"v+(val*g)*g": v + (val * g) * g, "v+val*(g*g)": v + val * (g * g), "fma(val*g,g,v)": fma(val * g, g, v),
             "fma(val,g*g,v)": fma(val, g * g, v), "fma(g*g... (g*val)": fma(g, g * val, v)}.items():
    print("addcmul", k, (bf16_round(c.astype(f)) != ref).sum())
x = bf16_round(rs.standard_normal(n).astype(f)); y = bf16_round(rs.standard_normal(n).astype(f))
t = bt(x).clone(); t.add_(bt(y), alpha=1 - 0.9); r = t.float().cpu().numpy()
A = f(1 - 0.9)
for k, c in {"x+A*y": x + A * y, "fma(A,y,x)": fma(A, y, x), "bf16A sep": x + bf16_round(np.array([A]))[0] * y}.items():
    print("add_alpha", k, (bf16_round(c.astype(f)) != r).sum())
c0 = bf16_round(rs.standard_normal(n).astype(f) * 1e-5); m = bf16_round(rs.standard_normal(n).astype(f) * 1e-3); cv = bf16_round(np.abs(rs.standard_normal(n)).astype(f) * 1e-3 + 1e-4)
t = bt(c0).clone(); t.addcdiv_(bt(m), bt(cv), value=-3.3e-6); r = t.float().cpu().numpy()
s = f(-3.3e-6)
for k, c in {"c+(s*m)/cv": c0 + (s * m) / cv, "c+s*(m/cv)": c0 + s * (m / cv), "fma(s,m/cv,c)": fma(s, m / cv, c0)}.items():
    print("addcdiv", k, (bf16_round(c.astype(f)) != r).sum())
t = bt(x).clone(); t.mul_(0.999); print("mul f32 scalar", (bf16_round(x * f(0.999)) != t.float().cpu().numpy()).sum())
vv = bf16_round(np.abs(rs.standard_normal(n)).astype(f) * 1e-6)
dc = (1 - 0.999 ** torch.tensor(3.0)) ** 0.5
sq = bt(vv).sqrt(); print("sqrt", (bf16_round(np.sqrt(vv)) != sq.float().cpu().numpy()).sum())
t = sq / dc; r1 = t.float().cpu().numpy()
print("div dc f32", (bf16_round(bf16_round(np.sqrt(vv)) / f(float(dc))) != r1).sum(), "mul by recip", (bf16_round(bf16_round(np.sqrt(vv)) * (f(1) / f(float(dc)))) != r1).sum())
t2 = t.clone().add_(1e-5, alpha=1); r2 = t2.float().cpu().numpy()
print("eps f32", (bf16_round(r1 + f(1e-5)) != r2).sum(), "eps bf16", (bf16_round(r1 + bf16_round(np.array([1e-5], dtype=f))[0]) != r2).sum())
p = bf16_round(rs.standard_normal(n).astype(f) * 0.02)
t = bt(p).clone(); t.mul_(1 - 1e-3 * 1e-2); print("wd mul", (bf16_round(p * f(1 - 1e-3 * 1e-2)) != t.float().cpu().numpy()).sum())
