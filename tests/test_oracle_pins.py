"""The oracle against every known-answer the reference leaves behind (notebook outputs):
per-model parameter counts and dataset patch counts.  These are the only machine-checkable pins the reference
offers (it has no tests and no golden vectors); numerics are "parity unpinned" (oracle/__init__.py)."""
import json
import os

import numpy as np

from oracle import models as M
from oracle import ops as O


def test_srcnn_param_count():
    L = M.srcnn_layers()
    assert M.count_params(L) == 28931                                   # SRCNN.ipynb:L141
    assert [int(np.prod(s)) + s[-1] for _, s in L] == [23424, 3104, 2403]


def test_edsr_param_count():
    L = M.edsr_layers(2, 3, 16, 64)
    assert M.count_params(L) == 1369859                                 # EDSR.ipynb:L392
    assert int(np.prod(L[-2][1])) + L[-2][1][-1] == 147712              # last up-conv
    assert int(np.prod(L[-1][1])) + L[-1][1][-1] == 1731                # out conv
    assert len(M.edsr_layers(4)) == len(L) + 1                          # x4: two up-convs, conv2d_36 is the output conv
    assert M.edsr_layers(4)[-1][0] == "conv2d_36"


def test_esrgan_generator_param_counts():
    L = M.esrgan_g_layers(2, 8, 4)
    assert M.count_params(L) == 1162915                                 # ESRGAN.ipynb:L636
    assert M.count_params(M.self_attention_layers("sa")) == 5232
    d = dict(L)
    per = [int(np.prod(d[f"rrdb_0_dense1_conv{k}"])) + d[f"rrdb_0_dense1_conv{k}"][-1] for k in range(1, 6)]
    assert per == [4616, 5192, 5768, 6344, 55360]
    assert M.count_params(M.esrgan_g_layers(4, 32, 23)) == 16930019


def test_discriminator_and_vgg_counts():
    assert M.count_params(M.discriminator_layers()) == 658305           # + 961 spectral-norm u vectors = 659266
    assert M.count_params(M.discriminator_layers()) + 961 == 659266     # ESRGAN.ipynb:L693-695
    assert M.count_params(M.vgg_base_layers(M.VGG19_CFG)) == 20024384   # ESRGAN.ipynb:L748
    assert M.count_params(M.vgg16_classifier_layers(2)) == 14846530     # VGG16.ipynb:L151-153
    head = M.vgg16_classifier_layers(2)[-2:]
    assert M.count_params(head) == 131842                               # the trainable part


def _count(h, w, p, s):
    return len(O.patch_positions(h + O.pad_amount(h, p, s), w + O.pad_amount(w, p, s), p, s))


def test_dataset_patch_counts():
    assert int(0.7 * 313 * _count(478, 478, 24, 12)) == 333251          # SRCNN.ipynb:L47
    assert 313 * _count(239, 239, 24, 12) == 112993                     # EDSR.ipynb:L46
    # defects loader pads but walks the UNPADDED size (loading_methods.py:275-277): 8x8 windows of 96/48 on 478
    assert int(0.7 * 313 * len(O.patch_positions(478, 478, 96, 48))) == 14022   # VGG16.ipynb:L73
    assert _count(512, 512, 48, 24) == 441 and _count(1080, 1920, 48, 24) == 45 * 80


def test_flop_model_matches_survey():
    """MAC/px figures bench.py's roofline uses."""
    macs = lambda L: sum(int(np.prod(s)) for _, s in L)
    assert macs(M.srcnn_layers()) == 28800
    dense = sum(int(np.prod(s)) for n, s in M.esrgan_g_layers(4, 32, 23) if n.startswith("rrdb_0_dense1_"))
    assert dense == 239616


# ---------------------------------------------------------------------------------------------------------------------
# Keras model.summary() tables the notebooks printed: every row's layer name, type, OUTPUT SHAPE and parameter count
# (tests/golden/notebook_summaries.json, extracted by tests/golden/make_pins.py) against the oracle's graph restatement.
# ---------------------------------------------------------------------------------------------------------------------
def _nb():
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "notebook_summaries.json")))


def _same(rows, pinned):
    assert len(rows) == len(pinned["rows"]), (len(rows), len(pinned["rows"]))
    for got, want in zip(rows, pinned["rows"]):
        assert list(got) == list(want), (got, want)
    assert sum(r[3] for r in rows) == pinned["totals"]["Total"]


def test_generator_summary_rows_match_the_notebook():
    nb = _nb()["Generator"]                                   # ESRGAN.ipynb cell 6: x2, G=8, NB=4 on 24x24 patches, 151 rows
    rows = M.keras_summary_generator(scale=2, growth=8, num_rrdb=4, h=24, w=24)
    _same(rows, nb)
    assert rows[-1][2] == [None, 48, 48, 3]
    # and the bench configuration is the same construction: x4 / G=32 / NB=23 ends at 4x the patch, 16 930 019 parameters
    big = M.keras_summary_generator(scale=4, growth=32, num_rrdb=23, h=48, w=48)
    assert big[-1][2] == [None, 192, 192, 3] and sum(r[3] for r in big) == 16930019
    assert {tuple(r[2]) for r in big if r[0].startswith("self_attention")} == {(None, 48, 48, 64), (None, 96, 96, 64)}


def test_discriminator_summary_rows_match_the_notebook():
    nb = _nb()["Discriminator"]                               # maps 48 -> 24 -> 24 -> 12 -> 12 -> 6 -> GAP (ESRGAN.ipynb:L693)
    rows = M.keras_summary_discriminator(48, 48)
    _same(rows, nb)
    assert nb["totals"]["Non-trainable"] == 961 == sum(f for f in (64, 64, 64, 128, 128, 256, 256, 1))
    assert [r[2][1] for r in rows if r[1] == "SpectralNormalization" and len(r[2]) == 4] == [48, 24, 24, 12, 12, 6]


def test_vgg19_extractor_and_vgg16_classifier_rows_match_the_notebooks():
    _same(M.keras_summary_vgg(M.VGG19_CFG, 48, 48, "input_1", last="block5_conv4"), _nb()["VGG_Feature_Extractor"])
    _same(M.keras_summary_vgg16_classifier(96, 96, 2), _nb()["vgg16_finetune"])     # VGG16.ipynb:L151: 96 -> 48 -> 24 -> 12 -> 6 -> 3


def test_srcnn_and_edsr_summary_rows_match_the_notebooks():
    _same(M.keras_summary_srcnn(24, 24), _nb()["sequential"])
    _same(M.keras_summary_edsr(2, 16, 64), _nb()["EDSR"])
    assert M.keras_summary_edsr(4)[-2][0] == "conv2d_36"      # x4: two up-convs (EDSR_model.py:85-90)
