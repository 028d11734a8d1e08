"""tests/golden/clip_pooler_tiny.npz: `pooler_output` of the REAL reference CLIPTextEncoder.forward
(/root/reference/model/text_encoder.py:340-354, fp16 on CPU) for the weights and token ids already pinned in
clip_text_tiny.npz -- the generator first checks that its `last_hidden_state` equals that fixture bit for bit.
Run in the build container only (imports /root/reference); commits data, never source."""
import os, sys, types
import numpy as np
import torch
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
for m in ("ftfy", "nltk", "h5py"):
    sys.modules.setdefault(m, types.ModuleType(m))
import model.text_encoder as te

z = np.load(os.path.join(ROOT, "tests", "golden", "clip_text_tiny.npz"))
sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("sd.")}
width = sd["ln_final.weight"].shape[0]
layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
enc = te.CLIPTextEncoder(sd["text_projection"].shape[1], sd["positional_embedding"].shape[0],
                         sd["token_embedding.weight"].shape[0], width, width // 64, layers)
te.convert_weights(enc)
enc.load_state_dict(sd)
enc.eval()
ids = torch.from_numpy(z["ids"].copy())
with torch.no_grad():
    out = enc(ids)
assert torch.equal(out["last_hidden_state"], torch.from_numpy(z["hidden"].copy())), "fixture mismatch"
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "clip_pooler_tiny.npz"), pooler_output=out["pooler_output"].numpy())
print("pooler_output", tuple(out["pooler_output"].shape), out["pooler_output"].dtype)
