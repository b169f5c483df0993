import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from hip_adapter import editnet_modules, load_numpy_state, to_dev
from oracle import cases
from show_edit_tell_amd import evaluate, editnet

# (1) wide greedy at 1..4 rows against the per-step loop
d, xe, rl = editnet_modules("editnet_full_b4")
for B in (1, 2, 3, 4):
    args = (d["wm"], to_dev(d["prev"][:B]), to_dev(d["plen"][:B]), to_dev(d["X"][:B]), True, False)
    with torch.no_grad():
        rl(*args); rl(*args)
        os.environ["SET_DEC_WIDE_MINB"] = "1"
        a = rl(*args)
        os.environ["SET_DEC_WIDE_MINB"] = "5"
        os.environ["SET_DEC_PERSISTENT"] = "0"
        b = rl(*args)
        del os.environ["SET_DEC_PERSISTENT"]
        torch.cuda.synchronize()
    print("wide greedy B=%d: ids equal %s, max dlogp %.2e" % (B, bool(torch.equal(a[0], b[0])), float((a[1] - b[1]).abs().max())))

# (2) beam: persistent vs batched
c, wm = d["case"], d["wm"]
for boost in (4.0, 2.0):
    sd = {k: v.copy() for k, v in d["sd"].items()}
    sd["fc.bias"][wm["<end>"]] += np.float32(boost)
    m = load_numpy_state(editnet.DecoderC(wm, c["D"], c["D"], c["D"], c["A"], c["F"]), sd)
    X, prev, plen = to_dev(d["X"]), to_dev(d["prev"]), to_dev(d["plen"])
    bs, bsc = evaluate.beam_search_editnet_batched(m, X, prev, plen, wm, 3, return_scores=True)
    for b in range(2):
        one = (X[b:b + 1], prev[b:b + 1], plen[b:b + 1])
        evaluate.beam_search_editnet(m, *one, wm, 3)
        got = evaluate._beam_search_editnet_persistent(m, *one, wm, 3)
        print("boost", boost, "img", b, "batched", bs[b], bsc[b], "persistent", got)
