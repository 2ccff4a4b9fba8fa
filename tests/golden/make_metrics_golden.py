#!/usr/bin/env python3
"""Golden vectors for the retrieval metrics: run the REFERENCE's `MetricsProcessor.compute_DPR_scores` and
`compute_DPR_scores_with_pos_ids` (src/metrics/metrics_processors.py:481-601) on synthetic retrieval records and store
inputs + outputs in tests/golden/metrics.json.

Run only in the build container (needs /root/reference):   python tests/golden/make_metrics_golden.py

The reference module imports packages this image lacks (easydict, wandb, evaluate, ...), none of which the two methods use
beyond `EasyDict` as an attribute dict, so the two function definitions are taken out of the file's AST and compiled on
their own with stand-ins for `EasyDict` / `tqdm`; nothing of the reference's text is written to the repository -- only the
records (data) and the numbers the reference computed for them.
"""
import ast
import json
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FLMR_REFERENCE_ROOT", "/root/reference")


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def reference_functions():
    path = os.path.join(REF, "src", "metrics", "metrics_processors.py")
    tree = ast.parse(open(path).read())
    wanted = {"compute_DPR_scores", "compute_DPR_scores_with_pos_ids"}
    fns = [n for cls in tree.body if isinstance(cls, ast.ClassDef) for n in cls.body
           if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {f.name for f in fns} == wanted
    ns = {"np": np, "EasyDict": AttrDict, "tqdm": lambda x: x}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    return ns["compute_DPR_scores"], ns["compute_DPR_scores_with_pos_ids"]


def main():
    rng = random.Random(7)
    vocab = ["river", "Paris", "copper", "violin", "glacier", "tiger", "Amazon", "basalt", "cello", "harbor"]
    Ks = [1, 5, 10, 20]
    records = []
    for q in range(14):
        answers = rng.sample(vocab, 3)
        gold = answers[0] if q % 3 else rng.choice(vocab)
        n = 20 if q % 5 else 7                      # short lists are padded by repeating the last element (FLMR_executor.py:864-871)
        passages = []
        for r in range(n):
            words = [rng.choice(["the", "a", "of", "stone", "blue", "old"]) for _ in range(6)]
            if rng.random() < 0.25:
                words.insert(2, rng.choice(answers).upper() if rng.random() < 0.5 else rng.choice(answers))
            if rng.random() < 0.1:
                words.append(gold)
            passages.append({"passage_index": q * 100 + r, "passage_id": f"p{q * 100 + r}", "content": " ".join(words),
                             "score": 30.0 - r + rng.random()})
        while len(passages) < max(Ks):
            passages.append(dict(passages[-1]))
        pos = [f"p{q * 100 + rng.randrange(25)}" for _ in range(2)]
        records.append({"question_id": q, "top_ranking_passages": passages, "answers": answers, "gold_answer": gold,
                        "pos_item_ids": pos})
    dpr, dpr_pos = reference_functions()
    log1 = AttrDict(metrics=AttrDict())
    dpr(None, AttrDict(), {"batch_retrieval_result": records, "Ks": Ks}, log1)   # (self, module, data_dict, log_dict)
    log2 = AttrDict(metrics=AttrDict())
    dpr_pos(None, AttrDict(field="pos_item_ids"), {"batch_retrieval_result": records, "Ks": Ks}, log2)
    no_answers = [{k: v for k, v in r.items() if k not in ("answers", "gold_answer")} for r in records]
    log3 = AttrDict(metrics=AttrDict())
    dpr(None, AttrDict(), {"batch_retrieval_result": no_answers, "Ks": Ks}, log3)
    with open(os.path.join(HERE, "metrics.json"), "w") as f:
        json.dump({"Ks": Ks, "records": records, "pseudo_relevance": dict(log1.metrics), "pos_ids": dict(log2.metrics),
                   "pseudo_relevance_without_answers": dict(log3.metrics)}, f)
    print(dict(log1.metrics), dict(log2.metrics), dict(log3.metrics))


if __name__ == "__main__":
    main()
