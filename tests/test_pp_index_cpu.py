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


@pytest.mark.parametrize("tail_mode", [0, 2])
def test_tail_partition_covers_every_k_tile_once(tail_mode):
    """Whole rounds + split-K tail (md_gemm_args.tail_ws): for the launch shapes of the 256-image step on the free chip and on the
    248 CUs an RCCL kernel leaves, and for a sweep of tile counts, the host model of the kernel's work partition visits every
    (output tile, k-tile) exactly once, a tail unit is always a workgroup's LAST item and whole iterations (two k-tiles) long, and
    the whole tiles are dealt evenly (whole rounds)."""
    cases = [(256, 16, 248), (256, 12, 248), (768, 12, 248), (576, 16, 256), (616, 16, 256), (704, 16, 248), (256, 44, 248), (1920, 16, 256),
             (325, 18, 256), (355, 8, 256)]
    cases += [(t, nk, cu) for t in range(250, 700, 37) for nk in (4, 12, 16) for cu in (256, 250, 248, 240)]
    took = 0
    for total, nk, cus in cases:
        G, tail_first, units, split, tail_nk = pm.plan_tail(total, nk, cus, tail_mode)
        seen = np.zeros((total, nk), dtype=np.int32)
        loads = []
        for b in range(G):
            items = pm.workgroup_items(b, G, total, nk, tail_first, units, split, tail_nk)
            for n, (item, k0, kn) in enumerate(items):
                assert kn % 2 == 0 and kn >= 2
                if kn != nk:
                    assert n == len(items) - 1 and item >= tail_first, "a tail unit is the workgroup's last item"
                seen[item, k0:k0 + kn] += 1
            loads.append(sum(kn for _, _, kn in items))
        assert (seen == 1).all(), (total, nk, cus, G, tail_first, units, split)
        if units:
            took += 1
            assert tail_first % G == 0 and units <= G and split >= 2 and split * tail_nk == nk
            assert max(loads) - min(loads) <= tail_nk, "whole rounds + at most one unit per workgroup"
    assert took >= 5
    # the north_star's per-rank shape under the CU hold: 256 tiles on 248 workgroups, K = 1024
    assert pm.plan_tail(256, 16, 248, 0)[2:4] == (32, 4)
    assert pm.plan_tail(576, 16, 256, 0)[2] == 0, "64 left-over tiles: the raw-tile traffic would cost more than the round it saves"
