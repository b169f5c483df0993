"""CPU: the request coalescer's host logic (show_edit_tell_amd/serving.py) — batch planning in arrival order, <pad> padding
to a common caption length, every caller gets exactly its rows, failures reach every future of the batch."""
import threading
import time

import pytest
import torch

from show_edit_tell_amd import serving


def test_plan_batches_keeps_arrival_order_and_row_cap():
    assert serving.plan_batches([4, 4, 4, 4], 16) == [[0, 1, 2, 3]]
    assert serving.plan_batches([4, 4, 4, 4, 1], 16) == [[0, 1, 2, 3], [4]]
    assert serving.plan_batches([10, 8, 2, 20, 3], 16) == [[0], [1, 2], [3], [4]]      # 20 > cap runs alone, nobody overtakes
    assert serving.plan_batches([], 16) == []


def test_pad_and_cat_pads_with_zero_columns():
    a = (torch.ones(2, 5, dtype=torch.long), torch.tensor([[5], [3]]), torch.full((2, 3), 1.0))
    b = (torch.full((3, 7), 2, dtype=torch.long), torch.tensor([7, 7, 1]), torch.full((3, 3), 2.0))
    (prev, plen, x), offs = serving.pad_and_cat([a, b])
    assert prev.shape == (5, 7) and offs == [0, 2, 5]
    assert (prev[:2, :5] == 1).all() and (prev[:2, 5:] == 0).all() and (prev[2:] == 2).all()
    assert plen.tolist() == [5, 3, 7, 7, 1] and x[:2].eq(1).all() and x[2:].eq(2).all()
    (prev4, _, _), _ = serving.pad_and_cat([a, b], t_multiple=4)        # rounded up: 7 -> 8 columns, the extra one is <pad>
    assert prev4.shape == (5, 8) and (prev4[:, 7] == 0).all() and torch.equal(prev4[:, :7], prev)
    (prev1, _, _), _ = serving.pad_and_cat([a], t_multiple=2)           # a lone request with an odd length is padded too
    assert prev1.shape == (2, 6)


def test_coalescer_hands_every_caller_its_rows():
    calls = []

    def decode(prev, plen, X):
        calls.append(prev.shape[0])
        time.sleep(0.02)                                   # while this "decode" runs, the other requests queue up
        return prev[:, :1] * 10 + plen.reshape(-1, 1), X.sum(1)

    with serving.RequestCoalescer(decode, max_rows=16) as co:
        outs = {}

        def caller(i):
            prev = torch.full((4, 3 + i), i + 1, dtype=torch.long)
            f = co.submit(prev, torch.full((4, 1), i), torch.full((4, 2), float(i)))
            outs[i] = f.result(timeout=10)
        ths = [threading.Thread(target=caller, args=(i,)) for i in range(9)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    for i in range(9):
        ids, s = outs[i]
        assert ids.shape == (4, 1) and (ids == (i + 1) * 10 + i).all() and (s == 2.0 * i).all()
    assert sum(calls) == 36 and max(calls) <= 16
    assert co.requests == 9 and co.batches == len(calls) < 9      # requests did share decodes


def test_coalescer_failure_reaches_every_future_and_close_refuses():
    def decode(prev, plen):
        raise ValueError("boom")
    co = serving.RequestCoalescer(decode, max_rows=8)
    f = co.submit(torch.zeros(2, 3, dtype=torch.long), torch.ones(2, dtype=torch.long))
    with pytest.raises(ValueError, match="boom"):
        f.result(timeout=10)
    co.close()
    with pytest.raises(RuntimeError):
        co.submit(torch.zeros(1, 1, dtype=torch.long), torch.ones(1, dtype=torch.long))
