"""Host logic of the two-level trailing update of nncf_amd/gptq.py: which columns must take every block's error
immediately (`near_range_end`).  Brute force: a column needs the near treatment iff some group that starts strictly
inside the super-block reads it (gptq.py:312-333 computes group parameters from weight_tensor[:, c:c+gs] with all
earlier blocks applied)."""
import pytest

from nncf_amd.gptq import SUPER_BLOCKS, near_range_end


@pytest.mark.parametrize("block", [128, 100, 50, 32])
@pytest.mark.parametrize("gs", [16, 32, 64, 96, 128, 192, 256, 384, 768, 1024, 1536, 2048, 4608])
def test_near_range_end_covers_every_straddling_group(gs, block):
    K = 9216 if 9216 % gs == 0 else gs * 3
    sup = SUPER_BLOCKS * block
    for s1 in range(0, K, sup):
        s2 = min(s1 + sup, K)
        need = s2
        for c in range(0, K, gs):
            if s1 < c < s2:  # parameters computed while errors of this super-block are pending
                need = max(need, min(c + gs, K))
        assert near_range_end(s1, s2, gs, K) == need, (gs, block, s1)


def test_standard_group_sizes_never_extend():
    for gs in (32, 64, 128, 256, 512, 1024, 2048):
        for s1 in range(0, 8192, 1024):
            assert near_range_end(s1, s1 + 1024, gs, 8192) == (s1 + 1024)
