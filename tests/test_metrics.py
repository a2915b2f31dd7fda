"""B-Norm / Penalty BLEU against the numbers the reference's own scorers give on its shipped OUTPUT/ files, and
sentence BLEU (NLTK method 2) against hand-computed cases."""
import math
import os

import pytest

import util
from fira_icse_amd import metrics


@pytest.mark.skipif(not util.has_reference(), reason="reference OUTPUT/ files not present (GPU box)")
@pytest.mark.parametrize("name,bnorm,penalty", [
    ("output_fira", 17.666052790472175, 0.13298690691792042),      # SURVEY.md Appendix C.4 (paper: 17.67 / 13.30)
    ("output_nngen", 9.16, None), ("output_codisum", 16.55, None),
])
def test_scores_reproduce_reference_scorers(name, bnorm, penalty):
    ref = open(os.path.join(util.REFERENCE, "OUTPUT", "ground_truth")).read().split("\n")
    path = os.path.join(util.REFERENCE, "OUTPUT", name)
    if not os.path.exists(path):
        pytest.skip("no %s" % name)
    hyp = open(path).read().split("\n")
    if hyp and hyp[-1] == "":
        hyp = hyp[:-1]
    b = metrics.bnorm_bleu(ref, hyp)
    assert abs(b - bnorm) < (1e-9 if name == "output_fira" else 6e-3)
    if penalty is not None:
        assert abs(metrics.penalty_bleu(ref, hyp) - penalty) < 1e-12


def test_bnorm_basic_properties():
    refs = ["fix null pointer in parser", "add unit test for Foo.bar()", "update readme"]
    assert abs(metrics.bnorm_bleu(refs, refs) - 100.0) < 1e-9
    worse = ["fix pointer", "add test", "remove readme file now"]
    assert 0 < metrics.bnorm_bleu(refs, worse) < 60
    with pytest.raises(ValueError):
        metrics.bnorm_bleu(refs, ["x", "", "y"])
    assert metrics.mteval_tokens("Foo.bar(1,2) a-b 3-4") == ["foo", ".", "bar", "(", "1,2", ")", "a-b", "3", "-", "4"]


def test_sentence_bleu_method2_hand_cases():
    ref = "the cat sat on the mat".split()
    assert abs(metrics.sentence_bleu_method2([ref], ref) - 1.0) < 1e-12
    assert metrics.sentence_bleu_method2([ref], []) == 0.0
    assert metrics.sentence_bleu_method2([ref], "dog runs".split()) == 0.0
    hyp = "the cat sat".split()                       # p1 = 3/3, p2 = (2+1)/(2+1), p3 = (1+1)/(1+1), p4 = (0+1)/(1+1)
    want = math.exp(1 - 6 / 3) * math.exp(0.25 * math.log(0.5))
    assert abs(metrics.sentence_bleu_method2([ref], hyp) - want) < 1e-12
    hyp = "the the the cat".split()                   # clipping: 'the' counts at most twice -> p1 = 3/4
    p = [3 / 4, (1 + 1) / (3 + 1), (0 + 1) / (2 + 1), (0 + 1) / (1 + 1)]
    want = math.exp(1 - 6 / 4) * math.exp(sum(0.25 * math.log(x) for x in p))
    assert abs(metrics.sentence_bleu_method2([ref], hyp) - want) < 1e-12
