"""The second-generation tiled CG kernels (cmfrec_amd/csrc/cg2_kernels.hpp: slots in use only, vectors redistributed through
LDS, 49-FMA Gramian product; CMFREC_HIP_CG2=1 -- not the default, see device.hpp) and their 6-slot-tile variant
(CMFREC_HIP_CG2_NT6=1, three wavefronts per SIMD) on the operator parity cases, and the nnz bins of a half-step on one / three
streams instead of the default two (CMFREC_HIP_BINS_PAR)."""
import numpy as np
import pytest

import test_gpu_operators as T

pytestmark = pytest.mark.gpu
DT = [np.float64, np.float32]


@pytest.fixture(params=["cg2", "cg2-nt6", "cg2-tinyall", "cg2-pf", "pf-only"])
def second_generation(request, monkeypatch):
    if request.param != "pf-only":
        monkeypatch.setenv("CMFREC_HIP_CG2", "1")
    if request.param in ("cg2-pf", "pf-only"):        # the 33 .. 64 bin with the next row's tile prefetched into LDS by DMA
        monkeypatch.setenv("CMFREC_HIP_CG2_PF", "1")
    if request.param == "cg2-nt6":
        monkeypatch.setenv("CMFREC_HIP_CG2_NT6", "1")
    if request.param == "cg2-tinyall":
        monkeypatch.setenv("CMFREC_HIP_CG2_TINY", "all")
    return request.param


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("implicit", [True, False])
@pytest.mark.parametrize("k", [50, 64, 9])
def test_every_slot_count(oracles, dtype, implicit, k, second_generation):
    T.test_every_slot_count(oracles, dtype, implicit, k)


@pytest.mark.parametrize("dtype", DT)
def test_row_kernels(oracles, dtype, second_generation):
    T.test_optimizeA_implicit(oracles, dtype, 50, "cg")
    T.test_optimizeA_explicit(oracles, dtype, 33, 2, "cg")
    T.test_two_rows_per_wave(oracles, dtype, True, 50)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("streams", ["1", "3"])
def test_bins_on_other_stream_counts(oracles, dtype, streams, monkeypatch):
    monkeypatch.setenv("CMFREC_HIP_BINS_PAR", streams)
    T.test_every_slot_count(oracles, dtype, True, 50)
    T.test_every_slot_count(oracles, dtype, False, 50)
