"""`__graft_entry__.smoke()` is what the driver runs on the GPU box before the bench: run it in the suite too, so a routing change that breaks one
of its assertions (round 4: its "headline kernel" launch had become too small for that kernel's gate) shows up here and not at round end."""
import pytest

pytestmark = pytest.mark.gpu


def test_graft_entry_smoke(capsys):
    import __graft_entry__ as g
    g.smoke()
    assert "smoke ok" in capsys.readouterr().out
