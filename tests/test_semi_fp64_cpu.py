"""The float64 referee of the student/teacher step (tests/_semi.py + tests/_fp64.py) on the reduced
model, CPU only: the CPU-oracle fp32 leg agrees with it -- the plumbing the GPU tier's full-size
referee test stands on."""
import copy

import pytest
import torch

from tests import _fp64, _semi, _small


@pytest.mark.parametrize('kind', ['nesie', 'saqe'])
def test_float64_referee_of_the_student_teacher_step_agrees_with_the_fp32_cpu_leg(oracle_kernels, kind):
    from tests.test_parity_gpu import _semi_pair
    model = _semi_pair(kind)
    model64 = _semi.as_double(model)      # (before any leg runs: the replay counters travel with the copy)
    book = {}
    want_l, want_g, want_p = _semi.semi_step(model, torch.device('cpu'), oracle_kernels, book=book)
    assert int(want_p['valid'].sum()) > 0
    ref_l, ref_g, ref_p = _semi.semi_step(model64, torch.device('cpu'), _fp64.Fp64Kernels(),
                                          dtype=torch.float64, book=book)
    # the referee's OWN pseudo labels (before the replay) are the fp32 leg's: same decisions
    assert torch.equal(ref_p['valid'], want_p['valid'])
    assert set(ref_l) == set(want_l)
    for k in want_l:
        a, b = float(ref_l[k].sum()), float(want_l[k].sum())
        assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (k, a, b)
    names = sorted(ref_g)
    assert set(want_g) == set(names)
    ref = torch.cat([ref_g[n].flatten().double() for n in names])
    got = torch.cat([want_g[n].flatten().double() for n in names])
    rel = float((got - ref).norm() / ref.norm())
    print(f'{kind}: fp32 CPU leg vs float64, flat gradient rel. L2 {rel:.3e}')
    assert rel < 5e-3, rel
