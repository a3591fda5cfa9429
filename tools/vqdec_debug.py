"""Debug print: structure decoder errors vs fixtures (GPU)."""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from tests import _golden as G
from tests.test_gpu_vqvae import _decoder, KEYS
for name in ["vqdec_d1280_30l", "vqdec_d1280_30l_L512"]:
    fx = G.load(name)
    m = _decoder(fx)
    sid = G.t(fx["sequence_id"]).cuda() if "sequence_id" in fx else None
    out = m.decode(G.t(fx["tokens"]).cuda(), attention_mask=G.t(fx["attention_mask"]).cuda(), sequence_id=sid)
    for k in KEYS:
        got = out[k].cpu().numpy()
        if k == "predicted_aligned_error" and "pae_stride" in fx:
            got = got[:, ::int(fx["pae_stride"]), ::int(fx["pae_stride"])]
        d = np.abs(got - fx[k])
        idx = np.unravel_index(d.argmax(), d.shape)
        print(name, k, "err", d.max(), "at", idx, "mean err", d.mean(), "absmax", np.abs(fx[k]).max(), "ref bf16 max", float(fx["ref_bf16_maxdiff_" + k]), "mean", float(fx["ref_bf16_meandiff_" + k]))
    a = out["tensor7_affine"].cpu().numpy(); w = fx["tensor7_affine"]
    print("  rot err", np.abs(a[..., :9] - w[..., :9]).max(), "trans err", np.abs(a[..., 9:] - w[..., 9:]).max())
