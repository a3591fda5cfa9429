import sys, torch
sys.path.insert(0, "/root/repo")
from esm_amd import synth
from esm_amd.esmc import ESMC
from oracle import trunk_ref
d, H, NL = 960, 15, 30
w = synth.esmc_weights(d, NL, seed=21)
tok = synth.synth_tokens([256, 131, 64, 249], seed=12)
ref = trunk_ref.esmc_forward(tok, w, H, NL)
valid = tok != 1
for small in (1024, 0):
    m = ESMC(d, H, NL); m.load_state_dict(w); m.transformer.small_tokens = small; m = m.cuda()
    out = m(tok.cuda())
    emb = out.embeddings.cpu()[valid]
    cos = torch.nn.functional.cosine_similarity(emb, ref["embeddings"][valid], dim=-1).min().item()
    rel = (emb - ref["embeddings"][valid]).norm().item() / ref["embeddings"][valid].norm().item()
    err = (out.sequence_logits.cpu()[valid] - ref["sequence_logits"][valid]).abs().max().item()
    hs = out.hidden_states.cpu()[:, valid]; hr = ref["hidden_states"][:, valid]
    hrel = ((hs - hr).norm(dim=(1, 2)) / hr.norm(dim=(1, 2)))
    print(f"small_tokens={small}: min cos {cos:.6f} rel-l2 {rel:.5f} logits max err {err:.4f}  hidden rel-l2 layer 0/14/29: {hrel[0]:.5f} {hrel[14]:.5f} {hrel[29]:.5f}")
