"""CPU proof of the index arithmetic of the persistent ping-pong GEMM (gemm_pp.hip) through its host-side model:
staging offsets + swizzles + fragment reads deliver exactly the (row, k) elements the MFMA operand layout needs, for
both operand layouts, interior and ragged (clamped) tiles; the epilogue's lane map covers each quadrant exactly once
with 8 consecutive columns per lane at the address the kernel computes."""
import numpy as np
import pytest

from tests import pp_index_model as pm


@pytest.mark.parametrize("kc", [1, 0])
@pytest.mark.parametrize("rows,r0", [(512, 256), (300, 256), (256, 0)])
def test_stage_and_fragment_reads(kc, rows, r0):
    # A strips: wr * 64 (64 rows, two fragments); B strips: wc * 32 (32 columns, one fragment)
    assert pm.check_operand(kc, rows, 128, r0, 64, [0, 64], 64) is None
    assert pm.check_operand(kc, rows, 128, r0, 0, [0, 32, 64, 96], 32) is None


def test_epilogue_lane_map():
    seen = np.zeros((256, 256), dtype=np.int32)
    for wr in range(2):
        for wc in range(4):
            for IH in range(2):
                for JH in range(2):
                    m = pm.epilogue_map(wr, wc, IH, JH)
                    for (i, pp), v in m.items():
                        for lane in range(64):
                            row = IH * 128 + wr * 64 + i * 32 + (lane & 31)           # EpiLane.coords + block offsets
                            col = JH * 128 + wc * 32 + pp * 16 + (lane >> 5) * 8
                            for e in range(8):
                                assert tuple(v[lane, e]) == (row, col + e), (wr, wc, IH, JH, i, pp, lane, e, v[lane, e])
                                seen[row, col + e] += 1
    assert (seen == 1).all()


def test_epilogue_row_run_layout():
    """quad_rows (bf16 epilogues): after the half-wave swap and the 4 x 4 transpose inside each quad, lane 4 q + k of store t holds
    the 8 consecutive columns 8 k .. 8 k + 7 of row rq + t, rq = (q / 8) * 32 + (q % 8) * 4 -- i.e. the four lanes of a quad write one
    64-byte run -- and the four stores of the eight waves cover the 256 x 256 tile exactly once."""
    seen = np.zeros((256, 256), dtype=np.int32)
    for wr in range(2):
        for wc in range(4):
            for IH in range(2):
                for JH in range(2):
                    T = pm.quad_rows_model(wr, wc, IH, JH)
                    for t in range(4):
                        for lane in range(64):
                            q, k = lane >> 2, lane & 3
                            row = IH * 128 + wr * 64 + (q >> 3) * 32 + (q & 7) * 4 + t        # EpiLane::quad_coords + t
                            col = JH * 128 + wc * 32 + 8 * k
                            assert T[t][lane] == [(row, col + e) for e in range(8)], (wr, wc, IH, JH, t, lane, T[t][lane])
                            for e in range(8):
                                seen[row, col + e] += 1
    assert (seen == 1).all()
