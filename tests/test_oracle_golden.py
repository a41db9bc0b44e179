"""CPU: the oracle restatement must reproduce the reference's golden vectors (CPU vs CPU)."""
import pytest
import torch

import oracle
import oracle.layers
from cases import CASES
from util import golden_results, golden_state_dict, load_npz, replay


class _NS:
    pass


NS = _NS()
for mod in (oracle.layers, oracle.heads):
    for k, v in vars(mod).items():
        if isinstance(v, type):
            setattr(NS, k, v)

INT_KEYS = {"classes", "num_instances", "n_out"}


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference(name):
    data = load_npz(name)
    m, res = replay(CASES[name], NS, data)
    gold = golden_results(data)
    assert set(res) == set(gold), (sorted(set(res) ^ set(gold)))
    for k, g in gold.items():
        r = res[k]
        assert r.shape == g.shape, (k, r.shape, g.shape)
        if k in INT_KEYS or k.startswith("assign"):
            assert torch.equal(r.long(), g.long()), k
        else:
            # CPU reductions re-associate with the thread count: scale atol by the tensor's magnitude
            atol = 5e-6 * max(1.0, float(g.abs().max()))
            torch.testing.assert_close(r, g, rtol=1e-4, atol=atol, msg=lambda s: f"{name}:{k}: {s}")
    after = golden_state_dict(data, "sd_after.")
    sd = m.state_dict()
    for k, g in after.items():
        torch.testing.assert_close(sd[k].float(), g.float(), rtol=1e-5, atol=1e-6, msg=lambda s: f"{name}:{k}: {s}")


def test_state_dict_keys_match_reference():
    """Key-for-key state_dict parity is what lets a reference checkpoint load."""
    for name in ("bifpn_3to7_train", "fpn_3to7_eval", "od_forward_eval", "semseg_forward_eval"):
        data = load_npz(name)
        torch.manual_seed(0)
        m = CASES[name].build(NS)
        gold = golden_state_dict(data)
        assert list(m.state_dict().keys()) == list(gold.keys())
        for k, v in m.state_dict().items():
            assert tuple(v.shape) == tuple(gold[k].shape), k
