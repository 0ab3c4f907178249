import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "multitemplatematching-python_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    # MTM_TEST_ORDER=reverse | shuffle:<seed>: the same suite in another order (results must not depend on what ran before -
    # round 5's uint16 defect only showed behind a particular prefix of the suite; tools/session_r06_evidence.sh runs both)
    order = os.environ.get("MTM_TEST_ORDER", "")
    if order == "reverse":
        items.reverse()
    elif order.startswith("shuffle:"):
        import random
        random.Random(int(order.split(":", 1)[1])).shuffle(items)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU (/dev/kfd) in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- GPU suite: every parity test starts from the shipped defaults and from poisoned memory ---------------------------
_FRESH = {}
_GPU_TEST_NO = [0]


def fresh_options():
    """{option: value} of a context created just now: what the library ships as its defaults (plus whatever environment
    switch tools/alt_modes.sh set for this run of the suite)."""
    if not _FRESH:
        from MTM import _lib
        c = _lib.Context(0)
        _FRESH.update(c.options())
        del c
    return dict(_FRESH)


@pytest.fixture(autouse=True)
def _defaults_and_poison(request):
    """Ahead of every -m gpu test:
    (1) the process-wide default context (what MTM.matchTemplates / findMatches / computeScoreMap use) must carry the options of
        a fresh context - round 5's suite left it in the reciprocal-normalisation mode after its 33rd test, and everything behind
        that tested a mode that no longer ships as the default;
    (2) mtm_debug_poison: a byte pattern (0xFF = NaN, 0x7F = huge, alternating from test to test) into every wave slot's
        scratch memory, every CU's LDS and the per-call work buffers of every live context.  The reference's functions are
        pure functions of their arguments (MTM/__init__.py:92); round 5's uint16 kernel was not - its results
        depended on what the 101 tests before it had left in scratch memory.  A test that passes here passes whatever ran
        before it.
    Behind the test the options are compared again, so that the test that leaks a setting is the one that fails."""
    if "gpu" not in request.keywords or not _has_gpu():
        yield
        return
    import build as mtm_build
    mtm_build.build()
    from MTM import _lib
    want = fresh_options()
    ctx = _lib.default_context()
    # (the hit capacity is the one option the library moves by itself: a call whose peaks overflow it grows it for the calls
    # that follow - capacity only, never results; it is put back here instead of being compared)
    adaptive = _lib.OPT_HIT_CAPACITY
    ctx.set_option(adaptive, want[adaptive])
    assert ctx.options() == want, "the default context does not carry the shipped defaults at the start of this test"
    _GPU_TEST_NO[0] += 1
    pattern = 0xFF if _GPU_TEST_NO[0] % 2 else 0x7F
    if os.environ.get("MTM_TEST_POISON", "1") != "0":
        for c in _lib.live_contexts():
            try:
                c.debug_poison(pattern, 7 if c is ctx else 4)
            except _lib.MtmError:
                if c is ctx:
                    raise                                # (a context a finished test left with a call in flight refuses)
    yield
    left = ctx.options()
    left[adaptive] = want[adaptive]
    if left != want:
        for opt, value in want.items():          # do not let one leak fail every test behind it
            ctx.set_option(opt, value)
    assert left == want, "this test left the default context with other options than it found"
