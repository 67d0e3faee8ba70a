import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package(); eng = pkg.H2Agg(0)
dev = torch.device("cuda:0")
n = 1 << 20
rng = np.random.Generator(np.random.PCG64(3))
k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 31] &= 0x1f
table = eng.bases_generate(torch.from_numpy(k).to(dev).data_ptr(), n)
u = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); u[:, 31] &= 0x1f
cases = {"uniform": u, "all equal": np.tile(u[0], (n, 1))}
m = np.tile(((21888242871839275222246405745257275088548364400416034343698204186575808495617 - 1).to_bytes(32, "little")), n)
cases["all r-1"] = np.frombuffer(m, np.uint8).reshape(n, 32).copy()
for name, arr in cases.items():
    d = torch.from_numpy(arr).to(dev)
    for glv in (-1, 1):
        eng.msm_configure_glv(glv)
        eng.g1_msm_device(table, d.data_ptr(), n)
        eng.profile_reset(); eng.profile_enable(True)
        for _ in range(3): eng.g1_msm_device(table, d.data_ptr(), n)
        eng.profile_enable(False)
        st = eng.profile_stages()
        print("%-10s %-5s " % (name, "glv" if glv == 1 else "plain") + " ".join("%s=%.2f" % (k.replace("msm_", ""), v[0] / 3) for k, v in st.items()), flush=True)
