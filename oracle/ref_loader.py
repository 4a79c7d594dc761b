"""Import the *unmodified* reference hot path from /root/reference.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): only tests/, the golden
generator and bench.py's cpu_baseline leg may import this module.  The
reference exists only in the build container; on the GPU box `available()`
is False and everything that needs it must skip.

What is loaded (nothing is copied): nhd/Matcher.py, nhd/Node.py,
nhd/CfgTopology.py, nhd/NHDCommon.py.  Their single missing import is
`colorlog`, satisfied by oracle/_shim.
"""
import contextlib
import io
import logging
import os
import sys
import types

REF_ROOT = os.environ.get("NHD_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shim")
_mods = None


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "nhd", "Matcher.py"))


def load():
    """Returns a namespace with Node, Matcher, CfgTopology & friends."""
    global _mods
    if _mods is not None:
        return _mods
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    for p in (REF_ROOT, _SHIM):
        if p not in sys.path:
            sys.path.insert(0, p)
    # NHDCommon.GetLogger (NHDCommon.py:20-39) only configures a logger that has no handler yet:
    # pre-attaching a NullHandler keeps the reference silent (its as-shipped level is INFO and the
    # hot path formats a log line per node).
    for name in ("nhd.Node", "nhd.Matcher", "nhd.CfgTopology", "nhd.NHDCommon"):
        lg = logging.getLogger(name)
        lg.addHandler(logging.NullHandler())
        lg.setLevel(logging.CRITICAL + 1)
        lg.propagate = False
    import nhd.Node as node_mod
    import nhd.Matcher as matcher_mod
    import nhd.CfgTopology as top_mod
    ns = types.SimpleNamespace(
        node_mod=node_mod, matcher_mod=matcher_mod, top_mod=top_mod,
        Node=node_mod.Node, Matcher=matcher_mod.Matcher,
        CfgTopology=top_mod.CfgTopology, ProcGroup=top_mod.ProcGroup,
        Core=top_mod.Core, GPU=top_mod.GPU, VLANInfo=top_mod.VLANInfo,
        NICCoreDirection=top_mod.NICCoreDirection, SMTSetting=top_mod.SMTSetting,
        NUMASetting=top_mod.NUMASetting, GpuType=top_mod.GpuType,
        TopologyMapType=top_mod.TopologyMapType)
    _mods = ns
    return ns


_parser_cls = None


def triad_parser():
    """The unmodified reference nhd.TriadCfgParser.TriadCfgParser (row f3 of SURVEY.md section 8).  Its two absent
    third-party imports, `libconf` and `magicattr`, are satisfied by the restatements under oracle/_shim."""
    global _parser_cls
    if _parser_cls is None:
        load()
        for name in ("nhd.TriadCfgParser", "nhd.CfgParser"):
            lg = logging.getLogger(name)
            lg.addHandler(logging.NullHandler())
            lg.setLevel(logging.CRITICAL + 1)
            lg.propagate = False
        for name in ("libconf", "magicattr"):              # an empty placeholder module left by another test
            if name in sys.modules and not hasattr(sys.modules[name], "loads" if name == "libconf" else "get"):
                del sys.modules[name]
                sys.modules.pop("nhd.TriadCfgParser", None)
        import nhd.TriadCfgParser as tp
        _parser_cls = tp.TriadCfgParser
    return _parser_cls


def config_to_topology(text):
    """TriadCfgParser(text, False).CfgToTopology(False) as the scheduler calls it (nhd/NHDScheduler.py:262-270)."""
    return triad_parser()(text, False).CfgToTopology(False)


class VirtualClock:
    """Replaces `time` inside nhd.Node so IsBusy (Node.py:847-850) is deterministic."""

    def __init__(self, t=1.0e6):
        self.t = float(t)

    def install(self):
        ref = load()
        ref.node_mod.time = types.SimpleNamespace(monotonic=lambda: self.t)
        return self


def find_node(nl, top):
    """Unmodified reference Matcher().FindNode with stdout swallowed (Matcher.py:329 prints)."""
    ref = load()
    m = _matcher()
    with contextlib.redirect_stdout(io.StringIO()):
        return m.FindNode(nl, top)


_m = None


def _matcher():
    global _m
    if _m is None:
        _m = load().Matcher()
    return _m
