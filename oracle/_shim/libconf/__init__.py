"""Stand-in for the third-party `libconf` package (deploy/requirements.txt: libconf>=2.0.0), which the reference's
nhd/TriadCfgParser.py:3 imports and which is neither installed here nor vendored in /root/reference.

TEST INFRASTRUCTURE ONLY (oracle/README.md).  Restates the published behaviour of libconf 2.0.x `loads` that the
reference relies on: libconfig 1.7 grammar; groups -> AttrDict (dict with attribute access, AttributeError on a
missing name, a repeated name overwrites), lists `( )` -> tuple, arrays `[ ]` -> list, integers / hex / 64-bit `L`
suffix -> int, floats -> float, true/false (any case) -> bool, adjacent string literals concatenate, `#`, `//`
and `/* */` comments.  `@include` is not supported.  The product reader (nhd_amd/csrc/wire_digest.cpp) is written
against the same grammar, independently; tests/test_wire_digest.py compares the two on the same texts."""
import re


class ConfigParseError(RuntimeError):
    pass


class AttrDict(dict):
    def __getattr__(self, attr):
        try:
            return self[attr]
        except KeyError:
            raise AttributeError("'AttrDict' object has no attribute %r" % attr)

    def __setattr__(self, attr, value):
        self[attr] = value


_SKIP = re.compile(r"(?:\s+|#[^\n]*|//[^\n]*|/\*.*?\*/)+", re.S)
_TOKENS = [
    ("float", re.compile(r"[-+]?(?:\d+)?\.\d*(?:[eE][-+]?\d+)?|[-+]?\d+(?:\.\d*)?[eE][-+]?\d+")),
    ("hex", re.compile(r"0[Xx][0-9A-Fa-f]+(?:LL?)?")),
    ("int", re.compile(r"[-+]?[0-9]+(?:LL?)?")),
    ("bool", re.compile(r"(?i)(?:true|false)\b")),
    ("name", re.compile(r"[A-Za-z\*][-A-Za-z0-9_\*]*")),
    ("string", re.compile(r'"(?:[^"\\]|\\.)*"', re.S)),
    ("punct", re.compile(r"[=:;,{}\[\]()]")),
]
_ESC = {"n": "\n", "r": "\r", "t": "\t", "f": "\f", "\\": "\\", '"': '"'}


def _unescape(body):
    out, i = [], 0
    while i < len(body):
        c = body[i]
        if c != "\\":
            out.append(c); i += 1; continue
        e = body[i + 1]
        if e == "x":
            out.append(chr(int(body[i + 2:i + 4], 16))); i += 4
        elif e in _ESC:
            out.append(_ESC[e]); i += 2
        else:
            raise ConfigParseError("unknown escape \\%s" % e)
    return "".join(out)


def _tokenize(text):
    pos, toks = 0, []
    while True:
        m = _SKIP.match(text, pos)
        if m:
            pos = m.end()
        if pos >= len(text):
            return toks
        if text.startswith("/*", pos):
            raise ConfigParseError("unterminated comment")
        for kind, rx in _TOKENS:
            m = rx.match(text, pos)
            if m:
                toks.append((kind, m.group(0)))
                pos = m.end()
                break
        else:
            raise ConfigParseError("unexpected character %r at offset %d" % (text[pos], pos))


class _Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def take(self):
        tok = self.peek()
        self.i += 1
        return tok

    def settings(self, top):
        out = AttrDict()
        while True:
            kind, val = self.peek()
            if kind is None:
                if not top:
                    raise ConfigParseError("unterminated group")
                return out
            if (kind, val) == ("punct", "}"):
                if top:
                    raise ConfigParseError("unbalanced '}'")
                return out
            if kind == "bool":              # `true` / `false` are not reserved as setting names by this reader either
                kind = "name"
            if kind != "name":
                raise ConfigParseError("setting name expected, got %r" % (val,))
            self.take()
            if self.take() not in (("punct", "="), ("punct", ":")):
                raise ConfigParseError("'=' or ':' expected after %s" % val)
            out[val] = self.value()
            if self.peek() in (("punct", ";"), ("punct", ",")):
                self.take()

    def value(self):
        kind, val = self.take()
        if (kind, val) == ("punct", "{"):
            g = self.settings(False)
            if self.take() != ("punct", "}"):
                raise ConfigParseError("'}' expected")
            return g
        if (kind, val) in (("punct", "["), ("punct", "(")):
            close = "]" if val == "[" else ")"
            items = []
            if self.peek() == ("punct", close):
                self.take()
            else:
                while True:
                    item = self.value()
                    if close == "]" and isinstance(item, (list, tuple, dict)):
                        raise ConfigParseError("arrays hold scalars only")
                    items.append(item)
                    sep = self.take()
                    if sep == ("punct", ","):
                        if self.peek() == ("punct", close):
                            self.take()
                            break
                        continue
                    if sep == ("punct", close):
                        break
                    raise ConfigParseError("',' or closing bracket expected")
            return items if close == "]" else tuple(items)
        if kind == "string":
            s = _unescape(val[1:-1])
            while self.peek()[0] == "string":
                s += _unescape(self.take()[1][1:-1])
            return s
        if kind == "float":
            return float(val)
        if kind == "hex":
            return int(val.rstrip("L"), 16)
        if kind == "int":
            return int(val.rstrip("L"))
        if kind == "bool":
            return val.lower() == "true"
        raise ConfigParseError("value expected, got %r" % (val,))


def loads(text):
    p = _Parser(_tokenize(text))
    cfg = p.settings(True)
    return cfg


def load(f):
    return loads(f.read())
