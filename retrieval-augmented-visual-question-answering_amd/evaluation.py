"""Executor-side glue after retrieval: Ranking -> per-question `top_ranking_passages` records and Recall@K / Precision@K.

Restates, for the build's own harness and for drop-in use by an executor:
  * the conversion loop of `FLMRExecutor.evaluate_outputs` (src/executors/FLMR_executor.py:852-895): ranked
    (pid, rank, score) triples -> [{passage_index, passage_id, content, score}], short lists padded by repeating the
    last element up to max(Ks);
  * `compute_DPR_scores` (src/metrics/metrics_processors.py:481-542): pseudo-relevance -- a passage counts when its
    lower-cased content contains any lower-cased answer string; recall@K = share of questions with >= 1 such passage
    in the top K, precision@K = mean share of such passages in the top K; the same for the gold answer;
  * `compute_DPR_scores_with_pos_ids` (:547-601): ground-truth relevance by passage id.
Pure Python on host data (strings / ids): nothing here belongs on the GPU.
"""
from typing import Dict, Iterable, List, Sequence


def ranking_to_batch_result(ranking_dict: Dict, question_ids: Sequence, passage_index2id: Dict[int, object],
                            passage_contents: Sequence[str], max_K: int, extra_fields: Dict = None) -> List[dict]:
    """One record per question, in `question_ids` order (FLMR_executor.py:852-895).  `extra_fields[qid]` (optional dict
    with e.g. answers / gold_answer / pos_item_ids) is merged into the record."""
    out = []
    for qid, ranked in zip(question_ids, ranking_dict.values()):
        if hasattr(ranked, "pids"):   # data.RankedList: the numpy rows, no (pid, rank, score) tuples built
            idxs, scores = ranked.pids.tolist(), ranked.scores.tolist()
        else:
            idxs = [int(entry[0]) for entry in ranked]
            scores = [float(entry[2]) for entry in ranked]
        if idxs and len(idxs) < max_K:  # "simply replicate the last element to avoid crash" (:864-871)
            pad = max_K - len(idxs)
            idxs += [idxs[-1]] * pad
            scores += [scores[-1]] * pad
        rec = {"question_id": qid,
               "top_ranking_passages": [{"passage_index": i, "passage_id": passage_index2id[i], "content": passage_contents[i],
                                         "score": s} for i, s in zip(idxs, scores)]}
        if extra_fields and qid in extra_fields:
            rec.update(extra_fields[qid])
        out.append(rec)
    return out


def recall_pseudo_relevance(batch_result: Iterable[dict], Ks: Sequence[int]) -> Dict[str, float]:
    """metrics_processors.py:481-542.  Records need `answers` (list[str]) and `gold_answer` (str)."""
    batch_result = list(batch_result)
    n = len(batch_result)
    sums = {name: [0.0] * len(Ks) for name in ("precision", "recall", "gold_precision", "gold_recall")}
    for rec in batch_result:
        if "answers" not in rec:
            return {}
        answers = [a.lower() for a in rec["answers"]]
        gold = rec["gold_answer"].lower()
        has_any = [any(a in p["content"].lower() for a in answers) for p in rec["top_ranking_passages"]]
        has_gold = [gold in p["content"].lower() for p in rec["top_ranking_passages"]]
        for j, K in enumerate(Ks):
            n_any, n_gold = sum(has_any[:K]), sum(has_gold[:K])
            sums["recall"][j] += 1.0 if n_any > 0 else 0.0
            sums["precision"][j] += n_any / K
            sums["gold_recall"][j] += 1.0 if n_gold > 0 else 0.0
            sums["gold_precision"][j] += n_gold / K
    return {f"{name}_at_{K}": vals[j] / n for name, vals in sums.items() for j, K in enumerate(Ks)} if n else {}


def recall_with_pos_ids(batch_result: Iterable[dict], Ks: Sequence[int], field: str = "pos_item_ids") -> Dict[str, float]:
    """metrics_processors.py:547-601.  Records need `field` (collection of relevant passage ids)."""
    batch_result = list(batch_result)
    n = len(batch_result)
    rec_sum, prec_sum = [0.0] * len(Ks), [0.0] * len(Ks)
    for rec in batch_result:
        positives = set(rec[field])
        hit = [1 if p["passage_id"] in positives else 0 for p in rec["top_ranking_passages"][: max(Ks)]]
        for j, K in enumerate(Ks):
            h = sum(hit[:K])
            rec_sum[j] += 1.0 if h > 0 else 0.0
            prec_sum[j] += h / K
    out = {}
    for j, K in enumerate(Ks):
        out[f"{field}_precision_at_{K}"] = prec_sum[j] / n if n else 0.0
        out[f"{field}_recall_at_{K}"] = rec_sum[j] / n if n else 0.0
    return out
