"""Debug print: structure decoder errors vs fixtures (GPU)."""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from tests import _golden as G
from tests.test_gpu_vqvae import _decoder, KEYS
for name in ["vqdec_tiny", "vqdec_tiny_seqid", "vqdec_d1280_30l"]:
    fx = G.load(name)
    m = _decoder(fx)
    sid = G.t(fx["sequence_id"]).cuda() if "sequence_id" in fx else None
    out = m.decode(G.t(fx["tokens"]).cuda(), attention_mask=G.t(fx["attention_mask"]).cuda(), sequence_id=sid)
    for k in KEYS:
        d = np.abs(out[k].cpu().numpy() - fx[k])
        idx = np.unravel_index(d.argmax(), d.shape)
        print(name, k, "err", d.max(), "at", idx, "mean err", d.mean(), "absmax", np.abs(fx[k]).max(), "ref bf16", float(fx["ref_bf16_maxdiff_" + k]))
    a = out["tensor7_affine"].cpu().numpy(); w = fx["tensor7_affine"]
    print("  rot err", np.abs(a[..., :9] - w[..., :9]).max(), "trans err", np.abs(a[..., 9:] - w[..., 9:]).max())
