"""ORACLE tooling (authoring container only): run the reference's OWN beam search to pin beam parity.

The reference's `evaluate()` (`editnet.py:595-740`, `dcnet.py:405-541`) and `evaluate_full()`
(`eval/eval xe/eval_full.py:88-240`) cannot be called as they stand: they need a DataLoader, tqdm,
the COCO scorers, and `top_k_words / vocab_size` produces a float index on torch >= 1.5
(SURVEY.md §8c.3).  This helper parses the reference file with `ast`, takes the body of the
per-image `for ... in enumerate(tqdm(loader ...))` loop up to (excluding) the sentence
construction, rewrites the single `top_k_words / vocab_size` true division into a floor division
(what the authors' torch 1.2 computed for integer tensors), wraps the statements into a function
of the loop variables and executes it against the reference's own model classes
(oracle/ref_slice.py).  No reference text is stored in this repository: the source is read from
/root/reference at run time, only numbers (token ids, scores) are written to tests/golden by
`python -m oracle.make_beam_golden`.
"""
from __future__ import annotations

import ast
import os

from . import ref_slice


class _FloorDivTopK(ast.NodeTransformer):
    """`top_k_words / vocab_size` -> `top_k_words // vocab_size` (the only `/` on an index tensor)."""

    def __init__(self):
        self.hits = 0

    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div) and isinstance(node.left, ast.Name) and node.left.id == "top_k_words":
            self.hits += 1
            return ast.copy_location(ast.BinOp(left=node.left, op=ast.FloorDiv(), right=node.right), node)
        return node


def _assigns_name(stmt, name):
    return isinstance(stmt, ast.Assign) and any(isinstance(t, ast.Name) and t.id == name for t in stmt.targets)


def load_beam_fn(relpath, func_name, extra_args):
    """Build `beam_one(<extra_args>, word_map, beam_size, <loop variables...>)` from the per-image loop body of
    `func_name` in REF_ROOT/relpath.  Returns (callable, loop variable names)."""
    import torch
    import torch.nn.functional as F
    path = os.path.join(ref_slice.REF_ROOT, relpath)
    with open(path, "r") as f:
        tree = ast.parse(f.read(), filename=path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == func_name)
    loop = next(n for n in fn.body if isinstance(n, ast.For))
    # loop target: (i, (a, b, c, ...))
    names = [e.id for e in loop.target.elts[1].elts]
    cut = next(i for i, s in enumerate(loop.body) if _assigns_name(s, "sen_idx"))
    body = loop.body[:cut]
    tr = _FloorDivTopK()
    body = [tr.visit(s) for s in body]
    assert tr.hits == 1, "expected exactly one `top_k_words / vocab_size` in %s:%s" % (relpath, func_name)
    ret = ast.parse("return seq, complete_seqs, complete_seqs_scores, infinite_pred").body[0]
    args = list(extra_args) + ["word_map", "beam_size"] + names
    fdef = ast.FunctionDef(
        name="beam_one",
        args=ast.arguments(posonlyargs=[], args=[ast.arg(arg=a) for a in args], kwonlyargs=[], kw_defaults=[], defaults=[]),
        body=[ast.parse("vocab_size = len(word_map)").body[0]] + body + [ret], decorator_list=[])
    mod = ast.fix_missing_locations(ast.Module(body=[fdef], type_ignores=[]))
    ns = dict(torch=torch, F=F, device=torch.device("cpu"))
    exec(compile(mod, path, "exec"), ns)
    return ns["beam_one"], names


def editnet_beam():
    """(decoder, word_map, beam_size, img, image_id, previous_caption, prev_caplen) -> reference editnet.py:603-713"""
    return load_beam_fn("editnet.py", "evaluate", ["decoder"])


def dcnet_beam():
    """(dae, word_map, beam_size, image_id, previous_caption, prev_caplen) -> reference dcnet.py:413-514"""
    return load_beam_fn("dcnet.py", "evaluate", ["dae"])


def ensemble_beam():
    """(dae_ar, decoder, word_map, beam_size, img, image_id, previous_caption, prev_caplen) -> eval_full.py:96-210"""
    return load_beam_fn(os.path.join("eval", "eval xe", "eval_full.py"), "evaluate_full", ["dae_ar", "decoder"])
