#!/usr/bin/env python3
"""Golden vectors for afcm_amd/predictor.py, captured from the reference's own functions.

models/predictor.py cannot be imported here (SimpleITK is absent), so the two pure functions the vectors need --
``remove_halo`` (models/predictor.py:17-51) and ``SliceBuilder._gen_indices`` (data/utils.py:118-124) -- are compiled on
their own from the reference's syntax tree at generation time and run on seeded inputs; only inputs and outputs are stored
(tests/golden/P1_predictor.npz).  Run in the build container: python tools/gen_golden_predictor.py
"""
import ast
import os

import numpy as np

REF = '/root/reference'


def extract(path, name):
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            node.decorator_list = []
            ns = {}
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
            return ns[name]
    raise KeyError(name)


def main():
    remove_halo = extract(os.path.join(REF, 'models/predictor.py'), 'remove_halo')
    gen_indices = extract(os.path.join(REF, 'data/utils.py'), '_gen_indices')
    rng = np.random.default_rng(0)
    out = {}
    # _gen_indices(i, k, s)
    iks = [(20, 8, 4), (21, 8, 4), (8, 8, 4), (37, 16, 12), (256, 256, 1), (30, 7, 7)]
    out['gi_args'] = np.array(iks, dtype=np.int64)
    for n, (i, k, s) in enumerate(iks):
        out[f'gi_{n}'] = np.array(list(gen_indices(i, k, s)), dtype=np.int64)
    # remove_halo(patch, index, shape, halo): every border / interior combination on a small volume
    shape = (12, 20, 24)
    cases = []
    for halo in ((1, 2, 3), (0, 2, 2), (2, 0, 4)):
        for z in (0, 4):
            for y in (0, 6, 12):
                for x in (0, 8, 16):
                    cases.append((halo, (z, z + 8), (y, y + 8), (x, x + 8)))
    meta = []
    for n, (halo, zz, yy, xx) in enumerate(cases):
        patch = rng.standard_normal((2, zz[1] - zz[0], yy[1] - yy[0], xx[1] - xx[0])).astype(np.float32)
        index = (slice(0, 2), slice(*zz), slice(*yy), slice(*xx))
        got, idx = remove_halo(patch, index, shape, halo)
        out[f'rh_patch_{n}'] = patch
        out[f'rh_out_{n}'] = np.ascontiguousarray(got)
        meta.append(list(halo) + list(zz) + list(yy) + list(xx) + [v for s in idx[1:] for v in (s.start, s.stop)])
    out['rh_meta'] = np.array(meta, dtype=np.int64)
    out['rh_shape'] = np.array(shape, dtype=np.int64)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'P1_predictor.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, len(cases), 'halo cases')


if __name__ == '__main__':
    main()
