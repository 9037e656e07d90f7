"""Static check of the generated gfx950 ISA (no GPU needed: hipcc cross-compiles): the write-after-read hazard behind round 2's
non-finite stem gradients must not be present in any kernel of the library (tools/isa_store_hazard.py explains it)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_128_bit_store_with_a_scalar_offset_is_overwritten_too_early():
    import isa_store_hazard
    assert isa_store_hazard.main() == 0
