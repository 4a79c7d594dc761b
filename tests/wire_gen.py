"""Random Triad pod configs in libconfig syntax (the wire format of SURVEY.md section 8 row f3) for the parity tests
of the wire digest.  The reference ships no sample config; the shapes below follow what nhd/TriadCfgParser.py
reads (134-309), with syntactic noise (comments, ':' / '=', optional separators, hex / 'L' integers, split strings)
and, on request, one defect that makes the reference return None or raise."""
import numpy as np


def _int(rng, v):
    r = rng.random()
    if v >= 0 and r < 0.1:
        return hex(v)
    if r < 0.2:
        return f"{v}L"
    return str(v)


def _num(rng, v):
    if isinstance(v, float):
        return repr(v)
    return _int(rng, v)


def _string(rng, s):
    if len(s) > 2 and rng.random() < 0.15:
        k = int(rng.integers(1, len(s)))
        return f'"{s[:k]}" "{s[k:]}"'
    return f'"{s}"'


def _seq(rng, items, brackets):
    tail = "," if items and rng.random() < 0.1 else ""
    return brackets[0] + ", ".join(items) + tail + brackets[1]


def _setting(rng, name, value):
    eq = " = " if rng.random() < 0.8 else " : "
    end = ";" if rng.random() < 0.85 else ("," if rng.random() < 0.5 else "")
    note = (" /* " + name + " */" if rng.random() < 0.5 else "  # " + name + "\n  ") if rng.random() < 0.05 else ""
    return f"{name}{eq}{value}{end}{note}\n"


def make_config(seed, defect=None, types_hi=4, inst_hi=3):
    """Returns the config text.  defect in {None, 'no_topology', 'no_cpu_arch', 'bad_arch', 'no_ext_smt', 'ext_missing',
    'no_map_type', 'no_module', 'two_numa_dp', 'nic_cores_len', 'no_hugepages', 'helper_missing', 'int_of_string',
    'speed_index', 'syntax', 'helper_smt_missing', 'no_mod_defs'}"""
    rng = np.random.default_rng(seed)
    speeds = [0, 5, 10, 25, 40, 2.5, 12.5, 10.0]
    out = []
    if defect != "no_hugepages":
        hp = int(rng.choice([0, 2, 4, 8, 16]))
        out.append(_setting(rng, "Hugepages_GB", _num(rng, hp if rng.random() < 0.8 else float(hp))))
    n_ext = int(rng.integers(0, 4))
    ext_names = [f"misc.c{i}" if rng.random() < 0.7 else f"misc.arr[{i}]" for i in range(n_ext)]
    misc_fields = "".join(_setting(rng, f"c{i}", _int(rng, -1)) for i in range(n_ext))
    misc_fields += _setting(rng, "arr", _seq(rng, [_int(rng, -1)] * max(n_ext, 1), "[]"))
    n_types = int(rng.integers(1, types_hi))            # (module types x instances = processing groups: up to 6 by default)
    mod_defs, sections = [], []
    for t in range(n_types):
        mname = f"Mod{chr(65 + t)}"
        md = _setting(rng, "module", _string(rng, mname))
        n_inst = int(rng.integers(1, inst_hi))
        has_helpers = rng.random() < 0.6
        has_dp = rng.random() < 0.7
        has_nic = rng.random() < 0.3 or not has_dp
        helper_names = []
        if has_helpers:
            helper_names = [("h%d" % k, rng.random() < 0.4) for k in range(int(rng.integers(1, 4)))]
            md += _setting(rng, "helper_cores", _seq(rng, [_string(rng, n) for n, _ in helper_names], "()"))
            if defect != "helper_smt_missing":
                md += _setting(rng, "helper_cores_smt", "true" if rng.random() < 0.5 else "false")
        if rng.random() < 0.5:
            md += _setting(rng, "data_vlan", '"vlan"')
        if has_dp:
            dp = _setting(rng, "name", '"dp"') + _setting(rng, "proc_cores_smt", str(bool(rng.random() < 0.5)).upper() if rng.random() < 0.2 else ("true" if rng.random() < 0.5 else "false"))
            if rng.random() < 0.5:
                dp += _setting(rng, "gpu_type", _string(rng, str(rng.choice(["V100", "ANY", "2080Ti", "H100"]))))
            md += _setting(rng, "dp_group", "{\n" + dp + "}")
        if has_nic:
            md += _setting(rng, "nic_cores", _seq(rng, ['"nrx"', '"nrs"', '"ntx"', '"nts"', "true" if rng.random() < 0.5 else "false"]
                                                   + (['"extra"'] if defect == "nic_cores_len" and t == 0 else []), "()"))
        mod_defs.append("{\n" + md + "}")
        insts = []
        for k in range(n_inst):
            f = _setting(rng, "module", _string(rng, f"inst{k}"))
            for hn, is_arr in helper_names:
                if defect == "helper_missing" and t == 0 and k == 0 and hn == helper_names[0][0]:
                    continue
                if is_arr:
                    f += _setting(rng, hn, _seq(rng, [_int(rng, -1)] * int(rng.integers(0, 4)), "[]"))
                else:
                    f += _setting(rng, hn, '"x"' if defect == "int_of_string" and t == 0 and k == 0 else _int(rng, -1))
            f += _setting(rng, "vlan", _int(rng, int(rng.integers(1, 4000))))
            if has_dp:
                npairs = int(rng.integers(0, 3))
                d = _setting(rng, "rx_cores", _seq(rng, [_int(rng, -1)] * npairs, "[]"))
                d += _setting(rng, "tx_cores", _seq(rng, [_int(rng, -1)] * npairs, "[]"))
                rs = [_num(rng, speeds[int(rng.integers(0, len(speeds)))]) for _ in range(npairs)]
                ts = [_num(rng, speeds[int(rng.integers(0, len(speeds)))]) for _ in range(npairs)]
                if defect == "speed_index" and t == 0 and k == 0 and npairs:
                    rs = rs[:-1]
                # libconfig arrays are homogeneous: mixed int / float speeds go into a list instead
                def seq(vals):
                    mixed = len({("." in v or "e" in v) for v in vals}) > 1
                    return _seq(rng, vals, "()" if mixed or rng.random() < 0.3 else "[]")
                d += _setting(rng, "rx_speeds", seq(rs))
                d += _setting(rng, "tx_speeds", seq(ts))
                if rng.random() < 0.6:
                    d += _setting(rng, "cpu_workers", _seq(rng, [_int(rng, -1)] * int(rng.integers(0, 4)), "[]"))
                ngm = int(rng.integers(0, 5))
                ndev = max(1, int(rng.integers(1, 4)))
                entries = []
                for _g in range(ngm):
                    if rng.random() < 0.1:
                        entries.append(_seq(rng, [_int(rng, -1)] * 3, "[]"))         # malformed entry: skipped by the reference
                    else:
                        entries.append(_seq(rng, [_int(rng, -1), _int(rng, int(rng.integers(0, ndev)))], "[]"))
                d += _setting(rng, "gpu_map", _seq(rng, entries, "()"))
                groups = ["{\n" + d + "}"]
                if defect == "two_numa_dp" and t == 0 and k == 0:
                    groups = groups * 2
                f += _setting(rng, "dp", _seq(rng, groups, "()"))
            if has_nic:
                nn = int(rng.integers(0, 3))
                f += _setting(rng, "nrx", _seq(rng, [_int(rng, -1)] * nn, "[]"))
                f += _setting(rng, "ntx", _seq(rng, [_int(rng, -1)] * nn, "[]"))
                f += _setting(rng, "nrs", _seq(rng, [_num(rng, int(rng.choice([0, 5, 10, 25])))] * nn, "[]"))
                f += _setting(rng, "nts", _seq(rng, [_num(rng, float(rng.choice([0, 2.5, 10])))] * nn, "[]"))
            insts.append("{\n" + f + "}")
        if not (defect == "no_module" and t == 0):
            sections.append(_setting(rng, mname, _seq(rng, insts, "()")))
    topo = ""
    if defect != "no_cpu_arch":
        topo += _setting(rng, "cpu_arch", _string(rng, "ZEN9" if defect == "bad_arch" else str(rng.choice(["ANY", "SKYLAKE", "ICE_LAKE", "HASWELL"]))))
    if defect != "no_map_type":
        topo += _setting(rng, "map_type", _string(rng, str(rng.choice(["NUMA", "PCI", "NUMA", "PCI", "NONE", "numa"]))))
    topo += _setting(rng, "kni_vlan", '"kni"')
    topo += _setting(rng, "ext_cores", _seq(rng, [_string(rng, n) for n in ext_names] + (['"misc.nope"'] if defect == "ext_missing" else []), "()"))
    if defect != "no_ext_smt":
        topo += _setting(rng, "ext_cores_smt", "true" if rng.random() < 0.5 else "false")
    if defect != "no_mod_defs":
        topo += _setting(rng, "mod_defs", _seq(rng, mod_defs, "()"))
    parts = [_setting(rng, "misc", "{\n" + misc_fields + "}")] + sections
    if defect != "no_topology":
        parts.append(_setting(rng, "TopologyCfg", "{\n" + topo + "}"))
    order = rng.permutation(len(parts))
    text = "# generated pod config\n" + "".join(out) + "/* sections */\n" + "".join(parts[i] for i in order)
    if defect == "syntax":
        text = text.replace("{", "{ = ", 1)
    return text


DEFECTS = [None, "no_topology", "no_cpu_arch", "bad_arch", "no_ext_smt", "ext_missing", "no_map_type", "no_module",
           "two_numa_dp", "nic_cores_len", "no_hugepages", "helper_missing", "int_of_string", "speed_index", "syntax",
           "helper_smt_missing", "no_mod_defs"]
