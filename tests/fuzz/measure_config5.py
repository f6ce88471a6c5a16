import sys, os
REPO="/root/repo"; sys.path[:0]=[REPO, os.path.join(REPO,"1d-burgers-equation-roms_amd")]
import numpy as np, torch, copy
from burgers_hip import rom, decoder
from oracle import burgers_ref as br
import bench
g = np.load(os.path.join(REPO,"tests/golden/ann_n5.npz"))
model = bench.ann_model(g)
X = np.linspace(0,100,512)
mu1a, mu2a = bench.mu_shard(2048, 1, 0)
idx = np.linspace(0, 2047, 16).astype(int)
mu1, mu2 = mu1a[idx], mu2a[idx]
nT = 40
res = rom.pod_ann_run(X, np.ones(512), mu1, mu2, 0.05, nT, g["U_p"], g["U_s"], model)
torch.cuda.synchronize()
Ws=[g[f"W{i}"] for i in range(6)]; bs=[g[f"b{i}"] for i in range(6)]
it = res.iters.cpu().numpy(); fl = res.flags.cpu().numpy()
for b in range(16):
    Uo, ito = br.pod_ann_prom(X, 0.05, nT, np.ones(512), mu1[b], 0.0, mu2[b], g["U_p"], g["U_s"], Ws, bs, return_iters=True)
    rel = np.linalg.norm(res.hist[b].cpu().numpy().T - Uo)/np.linalg.norm(Uo)
    print("ANN", b, "mu=(%.3f,%.4f)"%(mu1[b],mu2[b]), "rel %.2e"%rel, "cap hip", int((it[b]>=50).sum()), "cap oracle", int((ito>=50).sum()), "iters equal", np.array_equal(it[b], ito), "maxdiff", int(np.abs(it[b]-ito).max()))
# bf16 closure
ref = rom.pod_ann_run(X, np.ones(512), mu1[:4], mu2[:4], 0.05, 10, g["U_p"], g["U_s"], model)
low = rom.pod_ann_run(X, np.ones(512), mu1[:4], mu2[:4], 0.05, 10, g["U_p"], g["U_s"], model, ann_dtype=torch.bfloat16)
e = (low.hist-ref.hist).flatten(1).norm(dim=1)/ref.hist.flatten(1).norm(dim=1)
print("bf16 intrusive closure rel err per sample", e.cpu().numpy())
# RBF
gr = np.load(os.path.join(REPO,"tests/golden/rbf_n17.npz"))
cl = (gr["U_p"], gr["U_s"], gr["X_train"], gr["W_gaussian"], float(gr["eps_gaussian"]), gr["x_min"], gr["x_max"], gr["y_min"], gr["y_max"])
r = rom.pod_rbf_run(X, np.ones(512), mu1, mu2, 0.05, 30, *cl)
it = r.iters.cpu().numpy()
for b in range(0,16,3):
    Uo, ito = br.pod_rbf_prom(X, 0.05, 30, np.ones(512), mu1[b], 0.0, mu2[b], *cl, return_iters=True)
    rel = np.linalg.norm(r.hist[b].cpu().numpy().T - Uo)/np.linalg.norm(Uo)
    print("RBF", b, "rel %.2e"%rel, "cap hip", int((it[b]>=30).sum()), "cap oracle", int((ito>=30).sum()), "iters equal", np.array_equal(it[b], ito))
# decoder bf16
gd = np.load(os.path.join(REPO,"tests/golden/nonintrusive_decoder.npz"))
md = bench.decoder_model(gd)
U32 = decoder.predict_on_grid(mu1[:8], mu2[:8], 501, gd["U_modes"], copy.deepcopy(md), gd["mean"], gd["std"])
U16 = decoder.predict_on_grid(mu1[:8], mu2[:8], 501, gd["U_modes"], copy.deepcopy(md), gd["mean"], gd["std"], dtype=torch.bfloat16)
e = (U16-U32).flatten(1).norm(dim=1)/U32.flatten(1).norm(dim=1)
print("bf16 decoder rel err vs fp32 per sample", e.cpu().numpy())
