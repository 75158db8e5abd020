"""Host-side properties of the P1 launch shapes (no GPU): the workgroup renumbering that gives every XCD one contiguous eighth of the pieces
(csrc/kernels.hip xcd_contiguous) is a permutation of the ids for every grid size, and an XCD's ids -- the hardware deals ids to the eight
XCDs round robin -- map to one ascending run.  The GPU side is tests/test_parity_gpu.py::test_dense_workgroup_order_and_cap_do_not_change_the_samples."""
import re
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def xcd_contiguous(wg, nwg):
    per = nwg >> 3
    return (wg & 7) * per + (wg >> 3) if wg < (per << 3) else wg


def test_the_restatement_is_the_kernels_formula():
    src = open(os.path.join(ROOT, "bevy-signed-distance-mesh-generation_amd", "csrc", "kernels.hip")).read()
    body = src[src.index("uint32_t xcd_contiguous("):]
    body = body[:body.index("}") + 1]
    assert re.search(r"per = nwg >> 3;", body)
    assert re.search(r"wg < \(per << 3\)\) \? \(wg & 7u\) \* per \+ \(wg >> 3\) : wg;", body)


def test_renumbering_is_a_permutation_and_an_xcd_writes_one_ascending_run():
    for nwg in list(range(1, 70)) + [255, 256, 257, 1033, 8226, 131842]:
        out = [xcd_contiguous(w, nwg) for w in range(nwg)]
        assert sorted(out) == list(range(nwg)), nwg
        per = nwg >> 3
        for x in range(8):
            mine = [out[w] for w in range(x, per << 3, 8)]            # the ids XCD x receives, in dispatch order
            assert mine == list(range(x * per, (x + 1) * per)), (nwg, x)
        assert out[per << 3:] == list(range(per << 3, nwg))           # the remainder keeps its ids
