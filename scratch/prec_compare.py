import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from tests.helpers import load, fixture_model, rel_l2
from tests.test_gpu_parity import build_dynamics, DEV, FORWARD_FIXTURES
from oracle import egnn_oracle as orc
torch.set_grad_enabled(False)
for name in FORWARD_FIXTURES:
    fx = load(name)
    sd_np, _, _ = fixture_model(fx)
    xh = torch.from_numpy(fx["xh"]).to(DEV); nm = torch.from_numpy(fx["node_mask"]).to(DEV); em = torch.from_numpy(fx["edge_mask"]).to(DEV)
    B = xh.shape[0]
    res = []
    for prec in ("fp32", "bf16x3", "bf16x6", "fp16x3"):
        dyn = build_dynamics(sd_np, int(fx["hidden_nf"]), int(fx["n_layers"]))
        dyn.precision = prec
        out = dyn._forward(torch.full((B, 1), float(fx["t_values"][0]), device=DEV), xh, nm, em, None, None).cpu().numpy()
        ref = fx["out_t0"]
        res.append((rel_l2(out, ref), rel_l2(out[..., :3], ref[..., :3]), rel_l2(out[..., 3:], ref[..., 3:]), np.abs(out-ref).max()))
    print(f"{name:28s} fp32: all {res[0][0]:.2e} vel {res[0][1]:.2e} h {res[0][2]:.2e} max {res[0][3]:.2e} | bf16x3: all {res[1][0]:.2e} vel {res[1][1]:.2e} h {res[1][2]:.2e} max {res[1][3]:.2e} | bf16x6: all {res[2][0]:.2e} vel {res[2][1]:.2e} h {res[2][2]:.2e} max {res[2][3]:.2e} | fp16x3: all {res[3][0]:.2e} vel {res[3][1]:.2e} h {res[3][2]:.2e} max {res[3][3]:.2e}")
# timing at the headline shape
from hierdiff_amd.weights import synthetic_state_dict
sd_np = synthetic_state_dict(9, 0, 256, 6, 2, True, 0, 1.0)
xh, nm, em = orc.random_inputs([30] * 256, 8, 1)
xh, nm = xh.to(DEV), nm.to(DEV)
t = torch.full((256, 1), 0.5, device=DEV)
outs = {}
for prec in ("fp32", "bf16x3", "bf16x6", "fp16x3"):
    dyn = build_dynamics(sd_np, 256, 6); dyn.precision = prec
    topo = dyn.topology(nm, None, 256, 30)
    dyn.sync_weights()
    for _ in range(3): o = dyn.forward_with_topology(topo, t, xh, None, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): o = dyn.forward_with_topology(topo, t, xh, None, None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    outs[prec] = o.cpu().numpy()
    print(prec, f"{dt*1e3:.3f} ms/forward")
print("bf16x3 vs fp32 at headline shape: rel_l2", rel_l2(outs["bf16x3"], outs["fp32"]))
print("bf16x6 vs fp32 at headline shape: rel_l2", rel_l2(outs["bf16x6"], outs["fp32"]))
print("fp16x3 vs fp32 at headline shape: rel_l2", rel_l2(outs["fp16x3"], outs["fp32"]))
# against the float64 evaluation of the oracle on a slice of the same batch: which mode is how far from the exact value
cfg = orc.DynCfg(in_node_nf=9, context_node_nf=0, hidden_nf=256, n_layers=6, normalization_factor=10.0)
xs, nms, ems = orc.random_inputs([30] * 8, 8, 1)
with torch.no_grad():
    ref32 = orc.dynamics_forward(orc.as_torch_sd(sd_np), cfg, torch.full((8, 1), 0.5), xs, nms, ems, prefix="dynamics.egnn.").numpy()
    with orc.float64():
        ref64 = orc.dynamics_forward(orc.as_torch_sd(sd_np), cfg, torch.full((8, 1), 0.5), xs, nms, ems, prefix="dynamics.egnn.").numpy()
print("float32 oracle (torch CPU) vs float64: rel_l2", rel_l2(ref32, ref64))
for prec in ("fp32", "bf16x3", "bf16x6", "fp16x3"):
    dyn = build_dynamics(sd_np, 256, 6); dyn.precision = prec
    o = dyn._forward(torch.full((8, 1), 0.5, device=DEV), xs.to(DEV), nms.to(DEV), ems.to(DEV), None, None).cpu().numpy()
    print(prec, "vs float64 oracle (8 molecules, H=256, L=6): rel_l2", rel_l2(o, ref64))
