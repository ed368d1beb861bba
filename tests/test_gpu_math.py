"""Device arithmetic the parity contract rests on: sequences that are shorter than the compiler's but must give the same bits."""
import pytest

from linevis_amd import capi

pytestmark = pytest.mark.gpu


def test_shortened_rsqrt_gives_the_ieee_bits_for_every_float(hip_lib):
    """lv_rsqrt_shade (lv_device.h: the argument clamped into [2^-60, 2^60], then v_sqrt_f32 + one-ulp correction and v_rcp_f32 + three
    Newton steps) against lv_rsqrt_shade_reference -- the same clamp followed by the compiler's own 1.0f / sqrtf(x) -- for ALL 2^32
    arguments: zeros, denormals, infinities, NaNs, negatives included.  The CPU checker's normalizeShade states the same clamped rule
    (and counts the calls the clamp acts on: tests/test_oracle.py); a single differing bit would show up as a parity failure
    somewhere else, far from its cause."""
    ctx = capi.Context(0)
    bad, first = ctx.selftest_rsqrt()
    assert bad == 0, "%d arguments differ, e.g. bits 0x%08x" % (bad, first)
