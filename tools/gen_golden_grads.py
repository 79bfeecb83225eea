#!/usr/bin/env python3
"""Golden parameter gradients of the training objective, from the REFERENCE's own modules.

Build-container only.  Runs the reference ProteinBertModel (imported from /root/reference through
the stand-ins of tools/gen_golden.py) on the tiny-config batch of tests/golden/tiny_forward.npz,
forms the loss of CodonModel.training_step (pretraining.py:109-130: mean CE + 0.4 * structure CE)
and calls backward.  Writes tests/golden/tiny_grads.npz: the loss and every parameter gradient.

    python tools/gen_golden_grads.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402


def main():
    os.chdir(G.REF)
    sys.path.insert(0, G.REF)
    G.install_shims()
    from oracle import transcodon_ref as O
    from oracle.weights import seeded_state_dict
    g = np.load(os.path.join(G.OUT, "tiny_forward.npz"))
    sd = seeded_state_dict(O.TINY, seed=int(g["seed"]), perturb=True)
    m = G.ref_model(O.TINY, sd).train()
    out = m(torch.from_numpy(g["tokens"]), torch.from_numpy(g["species"]))
    ce = torch.nn.CrossEntropyLoss(reduction="mean", ignore_index=-100)        # pretraining.py:54-55
    loss = ce(out["logits"].view(-1, 10), torch.from_numpy(g["labels"]).view(-1))
    struct = torch.from_numpy(g["struct_labels"])
    if bool((struct != -100).any()):                                           # pretraining.py:122-130
        loss = loss + 0.4 * ce(out["struct_label_output"].view(-1, 3), struct.view(-1))
    loss.backward()
    grads = {k.replace(".", "__"): p.grad.numpy() for k, p in m.named_parameters() if p.grad is not None}
    np.savez_compressed(os.path.join(G.OUT, "tiny_grads.npz"), loss=loss.detach().numpy(), **grads)
    print("loss", float(loss), "tensors with a gradient:", len(grads),
          "without:", sorted(k for k, p in m.named_parameters() if p.grad is None)[:6], "...")


if __name__ == "__main__":
    main()
