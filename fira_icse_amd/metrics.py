"""Text scorers of the reference's evaluation protocol, re-implemented without nltk / sumeval.

* ``bnorm_bleu`` / ``penalty_bleu``: the corpus scores the paper reports (reference Metrics/Bleu-B-Norm.py:26-129,
  171-185 and Metrics/Bleu-Penalty.py:160-186): NIST-mteval style tokenisation, per-sentence BLEU-4 with add-one
  smoothing on the 2..4-gram precisions, brevity penalty ``min(0, 1 - (r+1)/(c+1))`` in log space; B-Norm is the plain
  mean x 100, Penalty-BLEU the reference-length-weighted mean.  On the reference's shipped ``OUTPUT/`` files they give
  17.666 / 0.13299 (checked in tests/test_metrics.py when the reference tree is present).
* ``sentence_bleu_method2``: what ``dev()`` uses for checkpoint selection (reference run_model.py:22,171):
  NLTK ``sentence_bleu`` with ``SmoothingFunction().method2``.
"""
from __future__ import annotations

import math
import re
import sys
from collections import Counter
from typing import Iterable, List, Sequence
from xml.sax.saxutils import unescape

_TINY = sys.float_info.min
_SPLIT = re.compile(r"\w+|[^\s\w]")
# punctuation that is always set off by spaces: every ASCII punctuation mark (and the blank) except ' , - .
_ALWAYS = re.compile("([" + re.escape("{|}~[\\]^_` !\"#$%&()*+:;<=>?@/") + "])")
_AFTER_NONDIGIT = re.compile(r"([^0-9])([.,])")
_BEFORE_NONDIGIT = re.compile(r"([.,])([^0-9])")
_DIGIT_DASH = re.compile(r"([0-9])(-)")


def split_punct(line: str) -> str:
    return " ".join(_SPLIT.findall(line))


def mteval_tokens(s: str) -> List[str]:
    s = s.replace("<skipped>", "").replace("-\n", "").replace("\n", " ")
    s = unescape(s, {"&quot;": '"'})
    s = (" %s " % s).lower()
    s = _ALWAYS.sub(r" \1 ", s)
    s = _AFTER_NONDIGIT.sub(r"\1 \2 ", s)
    s = _BEFORE_NONDIGIT.sub(r" \1 \2", s)
    s = _DIGIT_DASH.sub(r"\1 \2 ", s)
    return s.split()


def _ngrams(words: Sequence[str], n: int = 4) -> Counter:
    c = Counter()
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            c[tuple(words[i:i + k])] += 1
    return c


def smoothed_sentence_bleu(refs: Sequence[str], hyp: str, n: int = 4):
    """(BLEU in [0,1], effective reference length) of one segment; refs/hyp are raw strings."""
    ref_tok = [mteval_tokens(r) for r in refs]
    hyp_tok = mteval_tokens(hyp)
    max_ref = Counter()
    for r in ref_tok:
        for g, c in _ngrams(r, n).items():
            max_ref[g] = max(max_ref[g], c)
    ref_len = min(len(r) for r in ref_tok)
    guess = [max(len(hyp_tok) - k + 1, 0) for k in range(1, n + 1)]
    correct = [0] * n
    for g, c in _ngrams(hyp_tok, n).items():
        correct[len(g) - 1] += min(max_ref.get(g, 0), c)
    log_b = 0.0
    for k in range(n):
        add = 1 if k > 0 else 0
        log_b += math.log(correct[k] + add + _TINY) - math.log(guess[k] + add + _TINY)
    log_b /= n
    log_b += min(0.0, 1.0 - float(ref_len + 1) / (len(hyp_tok) + 1))
    return math.exp(log_b), ref_len


def _pairs(references: Iterable[str], predictions: Iterable[str]):
    """Pair lines by running index exactly as the reference scripts do: blank reference lines are dropped BEFORE
    numbering, predictions keep their line numbers, and a blank prediction is an error."""
    refs = [r.strip() for r in references if r.strip()]
    preds = list(predictions)
    for i, p in enumerate(preds):
        if not p.strip():
            raise ValueError("prediction %d is empty (the reference scorer raises here too)" % i)
    for i, r in enumerate(refs):
        if i < len(preds):
            yield split_punct(r.lower()), split_punct(preds[i].strip().lower())


def bnorm_bleu(references: Iterable[str], predictions: Iterable[str]) -> float:
    scores = [smoothed_sentence_bleu([r], p)[0] for r, p in _pairs(references, predictions)]
    return 100.0 * sum(scores) / len(scores)


def penalty_bleu(references: Iterable[str], predictions: Iterable[str]) -> float:
    res = [smoothed_sentence_bleu([r], p) for r, p in _pairs(references, predictions)]
    total = float(sum(l for _, l in res))
    return sum(l / total * s for s, l in res)


def sentence_bleu_method2(references: Sequence[Sequence[str]], hypothesis: Sequence[str]) -> float:
    """NLTK ``sentence_bleu(references, hypothesis, smoothing_function=SmoothingFunction().method2)`` on token lists:
    uniform 4-gram weights, clipped precisions with +1/+1 smoothing on orders 2..4, closest-reference-length
    brevity penalty, 0 when no unigram matches or the hypothesis is empty."""
    hyp = list(hypothesis)
    if not hyp:
        return 0.0
    nums, dens = [], []
    for n in range(1, 5):
        counts = Counter(tuple(hyp[i:i + n]) for i in range(len(hyp) - n + 1))
        max_ref = Counter()
        for ref in references:
            rc = Counter(tuple(ref[i:i + n]) for i in range(len(ref) - n + 1))
            for g in counts:
                max_ref[g] = max(max_ref[g], rc[g])
        nums.append(sum(min(c, max_ref[g]) for g, c in counts.items()))
        dens.append(max(1, sum(counts.values())))
    if nums[0] == 0:
        return 0.0
    hyp_len = len(hyp)
    ref_len = min((abs(len(r) - hyp_len), len(r)) for r in references)[1]
    bp = 1.0 if hyp_len > ref_len else math.exp(1 - ref_len / hyp_len)
    logs = [math.log(nums[0] / dens[0])] + [math.log((nums[i] + 1) / (dens[i] + 1)) for i in range(1, 4)]
    return bp * math.exp(math.fsum(0.25 * x for x in logs))
