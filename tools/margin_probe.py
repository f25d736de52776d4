"""Median logit margin (logit[1] - logit[0]) of the hash-weight model on a golden case, from the oracle.
Used once per case to choose the node_classifier bias shift in oracle/cases.py (_BIAS_SHIFT)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import cases  # noqa: E402
from oracle import mdt_ref_cpu as R  # noqa: E402
from oracle import structure as S  # noqa: E402

kind = sys.argv[1]
hp = cases.real_hparams(kind)
trees = cases.real_trees(kind, hp)
t0 = time.time()
W = R.make_weights(hp, requires_grad=False)
print("weights", round(time.time() - t0, 1), "s")
batch = R.to_torch_batch(S.collate(trees, 5))
t0 = time.time()
with torch.no_grad():
    lo, _ = R.model_forward(W, hp, batch)
print("forward", round(time.time() - t0, 1), "s")
m = (lo[:, 1] - lo[:, 0]).numpy()
b = W["node_classifier.bias"].numpy()
print("bias", b, "margin median", float(np.median(m)), "min", m.min(), "max", m.max())
print("sorted margins", np.sort(m))
# shift s: bias := [+s/2, -s/2] REPLACES the hash bias; new margin = m - (b1 - b0) - s
m0 = m - (b[1] - b[0])
print("median of bias-free margin (use as _BIAS_SHIFT):", float(np.median(m0)))
ms = np.sort(m0)
n = len(ms)
lo_i, hi_i = n // 3, 2 * n // 3
gaps = ms[lo_i + 1:hi_i + 1] - ms[lo_i:hi_i]
k = int(np.argmax(gaps)) + lo_i
print(f"widest central gap: {ms[k]:.6f} .. {ms[k+1]:.6f} (width {ms[k+1]-ms[k]:.4f}); shift = {round(float((ms[k]+ms[k+1])/2), 4)}; "
      f"{int((m0 > (ms[k]+ms[k+1])/2).sum())} of {n} predicted positive")
