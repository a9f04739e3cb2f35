"""CPU tier: the host-side mirror of the reference API (validation, broadcasting, early exits, padding, dtype
rules, composition chains), with the three native primitives served by the CPU oracle, must reproduce every
golden case the reference produced -- bit for bit, values and masks."""
import pytest
import torch

from conftest import golden_ids
import case_runner

ALL = [c for c in golden_ids() if not c.endswith("cfg1_inputs") and not c.startswith("grads.")]   # (gradients: GPU tier, the oracle is forward-only)


@pytest.mark.parametrize("cid", ALL)
def test_golden_case_cpu(cid, golden, oracle_native):
    case = golden.cases[cid]
    got = case_runner.run_case(case, golden, torch.device('cpu'))
    case_runner.check_case(case, golden, got, exact_values=True)
