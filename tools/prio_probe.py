"""Single caller with prologue-ahead: does a stream priority for the caller's loop / the side stream change the rate?  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from show_edit_tell_amd import editnet_rl, synth
from show_edit_tell_amd.pipeline import DevicePrefetcher
B, R, F, T, V, D, A = 128, 36, 2048, 20, 10000, 1024, 512
dev = torch.device("cuda", 0)
wm = synth.word_map(V)
dec = editnet_rl.DecoderC(wm, D, D, D, A, F)
sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); dec = dec.to(dev).eval()
X = torch.from_numpy(synth.features(25, B, R, F)).to(dev)
prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(25, B, T, V, 5))
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)


def masked_stream(n_cus, first=0):
    """a HIP stream confined to CUs [first, first + n_cus) (hipExtStreamCreateWithCUMask), as a torch stream"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    words = 8                                            # 256 CUs
    mask = (ctypes.c_uint32 * words)()
    for cu in range(first, first + n_cus):
        mask[cu // 32] |= 1 << (cu % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def run(k, caller_prio, side_prio, whole=False, side=None):
    ahead = (lambda b: dec.decode_ahead(wm, b[1], b[2], b[0])) if whole else (lambda b: dec.begin_ahead(b[1], b[2], b[0]))
    pf = DevicePrefetcher(((X, prev, plen) for _ in range(k)), dev, depth=3 if whole else 2, streams=3 if whole else 1,
                          begin_ahead=ahead, stream_priority=side_prio, side_streams=side)
    cs = torch.cuda.current_stream(dev) if caller_prio is None else torch.cuda.Stream(dev, priority=caller_prio)
    cs.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(cs):
        for b in pf:
            out = dec(wm, b[1], b[2], b[0], True, False)
    torch.cuda.current_stream(dev).wait_stream(cs)
    return out


with torch.no_grad():
    for _ in range(3):
        dec(wm, prev, plen, X, True, False)
    for cp, sp in ((None, None), (-1, None), (None, 0), (-1, 0), (None, None), (-1, None)):
        try:
            run(5, cp, sp)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run(40, cp, sp)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print("caller prio %s side prio %s: %.0f decode-steps/s" % (cp, sp, 40 * 19 / dt), flush=True)
        except Exception as e:
            print("caller prio %s side prio %s: error %r" % (cp, sp, e))
    # the prologue-ahead side stream confined to a share of the CUs: its chip-filling GEMMs then leave the rest to the loop
    for n in (256, 128, 96, 64, 48):
        try:
            ms = masked_stream(n, 256 - n)
            run(5, None, None, side=[ms])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run(40, None, None, side=[ms])
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print("side stream on %3d CUs: %.0f decode-steps/s" % (n, 40 * 19 / dt), flush=True)
        except Exception as e:
            print("side stream on %d CUs: error %r" % (n, e))
