"""ORACLE tooling (authoring container only): beam-search golden vectors from the reference's own loops.

    python -m oracle.make_beam_golden        # rewrites tests/golden/beam_*.npz

For every case in oracle/cases.BEAM_CASES and every image of its batch, runs the per-image loop body of
the reference's `evaluate()` (EditNet `editnet.py:603-713`, DCNet `dcnet.py:413-514`) and of the
ensemble `evaluate_full()` (`eval/eval xe/eval_full.py:96-210`) — sliced and `/`->`//` patched by
oracle/ref_beam.py — on the reference's own model classes with the case's synthetic weights, and stores
numbers only: the chosen sequence, its score, how many hypotheses completed, whether the 50-step limit
was hit, and the score margin between the two best completed hypotheses (tests skip near-ties).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import cases, ref_beam, ref_slice  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
LMAX = 64


def _record(res):
    seq, comp, comp_scores, infinite = res
    seq = [int(w) for w in seq]
    sc = sorted((float(s) for s in comp_scores), reverse=True)
    return dict(seq=np.asarray(seq + [-1] * (LMAX - len(seq)), np.int64), n=np.int64(len(seq)),
                score=np.float64(sc[0] if (sc and not infinite) else np.nan), ncomplete=np.int64(len(comp)),
                infinite=np.bool_(bool(infinite)),
                margin=np.float64(sc[0] - sc[1] if len(sc) > 1 else np.inf))


def make(name):
    d = cases.build_beam(name)
    c, dc, wm = d["case"], d["dcase"], d["wm"]
    dec = ref_slice.load_state(ref_slice.editnet_xe()["DecoderC"](wm, c["D"], c["D"], c["D"], c["A"], c["F"]),
                               d["sd_e"]).eval()
    dae = ref_slice.load_state(ref_slice.dcnet_xe()["DAE"](wm, None, dc["D"], dc["A"], dc["C"], dc["E"]),
                               d["sd_d"]).eval()
    f_e, _ = ref_beam.editnet_beam()
    f_d, _ = ref_beam.dcnet_beam()
    f_x, _ = ref_beam.ensemble_beam()
    wrap = types.SimpleNamespace(dae=dae)               # evaluate_full() reaches the DAE as `dae_ar.dae`
    T_ = torch.from_numpy
    out = {}
    with torch.no_grad():
        for k in d["beams"]:
            recs = {"editnet": [], "dcnet": [], "ensemble": []}
            for b in range(c["B"]):
                img, prev, plen = T_(d["X"][b:b + 1]), T_(d["prev"][b:b + 1]), T_(d["plen"][b:b + 1])
                iid = torch.tensor([[b]])
                recs["editnet"].append(_record(f_e(dec, wm, k, img, iid, prev, plen)))
                recs["dcnet"].append(_record(f_d(dae, wm, k, iid, prev, plen)))
                recs["ensemble"].append(_record(f_x(wrap, dec, wm, k, img, iid, prev, plen)))
            for model, rs in recs.items():
                for field in rs[0]:
                    out["k%d.%s.%s" % (k, model, field)] = np.stack([r[field] for r in rs])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    summ = {m: "%d/%d finished" % (int((~out["k%d.%s.infinite" % (d["beams"][-1], m)]).sum()), c["B"])
            for m in ("editnet", "dcnet", "ensemble")}
    print("%-16s %6.1f KiB  %s" % (name, os.path.getsize(os.path.join(OUT, name + ".npz")) / 1024, summ))


def main(argv):
    assert ref_slice.have_reference(), "needs /root/reference (authoring container only)"
    torch.manual_seed(0)
    want = set(argv[1:])
    for name in cases.BEAM_CASES:
        if not want or name in want:
            make(name)


if __name__ == "__main__":
    main(sys.argv)
