"""Product-side parameter layout and initialisation against the reference's golden key list and seed-333 digests."""
import json
import math
import os

import torch

from mlperf_deepcam_amd import spec
from util_inputs import digest


def test_layout_matches_reference_state_dict(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "state_keys.json")))
    lay = spec.Layout()
    assert lay.state_keys == [k for k, _, _ in g["state_dict"]]
    assert list(lay.params) == g["parameters"]
    shapes = {k: s for k, s, _ in g["state_dict"]}
    for p in lay.params.values():
        assert list(p.shape) == shapes[p.name], p.name
    assert lay.n_params == 56454720 and len(lay.params) == 301
    offs = lay.offsets()
    assert offs[0] == 0 and offs[-1] == lay.n_params and all(b > a for a, b in zip(offs, offs[1:]))


def test_init_arena_bit_exact(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "init_seed333.json")))
    lay = spec.Layout()
    arena = torch.empty(lay.n_params)
    spec.init_arena(lay, arena, 333)
    for p in lay.params.values():
        d = digest(arena[p.offset:p.offset + math.prod(p.shape)])
        assert d["head"] == g[p.name]["head"] and d["sum"] == g[p.name]["sum"], p.name


def test_block_tables():
    b = {x.name: x for x in spec.blocks()}
    assert [s.stride for s in b["block1"].seps] == [1, 1, 2] and [bool(s.bn) for s in b["block1"].seps] == [True, True, False]
    assert [s.relu_after for s in b["block1"].seps] == [True, False, False]
    assert [(s.cin, s.cout) for s in b["block20"].seps] == [(728, 728), (728, 1024), (1024, 1024)]
    assert [s.relu_after for s in b["block20"].seps] == [True, False, False] and not b["block20"].relu_out
    assert all(s.relu_after for s in b["block7"].seps[:2]) and not b["block7"].seps[2].relu_after and not b["block7"].skip
