"""Pin the patch-gather index map to a REAL dependency of the reference: einops (reference requirements.txt:33 pins 0.4.1; 0.8.2 is
installed here -- `rearrange` patterns are stable across those).  vit-pytorch 0.33.2's `to_patch_embedding[0]` is exactly
`Rearrange('b c (h p1) (w p2) -> b (h w) (p1 p2 c)', p1=ph, p2=pw)` applied to the (B, C, 1, L) image the reference builds at
ecg_transformer/models/ecg_vit.py:141 with patch_size=(1, P) (:104).  Writes tests/golden/patch_gather_einops.npz: for an integer
ramp input, the gathered tokens produced by einops itself (not by the oracle's stand-in).

    python oracle/make_golden_einops.py            (this container only; the fixture is what travels)
"""
import os

import einops
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
    out = {'einops_version': np.array(einops.__version__)}
    for (L, P) in ((2560, 64), (5000, 20), (40, 10), (5000, 10)):
        x = np.arange(2 * 12 * L, dtype=np.int32).reshape(2, 12, L)
        img = x[:, :, None, :]                                                     # x.unsqueeze(-2): (B, C, 1, L)
        tok = einops.rearrange(img, 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)', p1=1, p2=P)
        out[f'L{L}_P{P}'] = np.ascontiguousarray(tok)
    np.savez_compressed(os.path.join(OUT, 'patch_gather_einops.npz'), **out)
    print('wrote', os.path.join(OUT, 'patch_gather_einops.npz'))


if __name__ == '__main__':
    main()
