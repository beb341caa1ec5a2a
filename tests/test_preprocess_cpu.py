"""CPU: the restatement of Pillow's bicubic resample (oracle/pil_resize_ref.py) against Pillow itself, and the product's
vectorised table builder against the restatement's."""
import numpy as np
import pytest
from PIL import Image

from oracle import pil_resize_ref as R

CASES = [(300, 400, 224), (100, 60, 224), (480, 640, 384), (224, 224, 224), (225, 223, 224), (37, 500, 224), (2, 3, 224),
         (1000, 17, 384), (224, 100, 224), (1, 1, 224)]


@pytest.mark.parametrize("h,w,S", CASES)
def test_restatement_equals_pillow(h, w, S):
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((S, S), resample=Image.BICUBIC))
    assert np.array_equal(R.resize_bicubic(img, S, S), ref)


def test_product_tables_equal_the_restatement():
    from embodied_captioning_amd.preprocess import pil_bicubic_coeffs
    for n_in in list(range(1, 70)) + [100, 223, 224, 225, 383, 384, 385, 480, 640, 1000, 1919]:
        for n_out in (224, 384):
            b0, k0 = R.coeffs(n_in, n_out)
            b1, k1 = pil_bicubic_coeffs(n_in, n_out)
            assert np.array_equal(b0, b1), (n_in, n_out)
            assert np.array_equal(k0, k1), (n_in, n_out)
