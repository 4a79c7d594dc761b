"""Row f3, the third-party layer: `oracle/_shim/libconf` and `oracle/_shim/magicattr.py` stand in for the two packages the
reference parser imports (libconf >= 2.0.0, magicattr >= 0.1.4 - absent here, no index access).  They cannot be pinned
against the real wheels; what CAN be pinned without them is the published behaviour: the example configuration of the
libconfig manual (chapter "Configuration Files": the `application` / `window` / `list` / `books` / `misc` file, reproduced
below from the manual) with the value types the manual and libconf's README assign to it (groups -> dict with attribute
access, lists -> tuple, arrays -> list, `L` suffix and hex -> int, adjacent strings concatenated, all three comment styles),
and magicattr's documented `get` (attribute chains and constant subscripts).  f3's parity stays labelled "third-party layer
unpinned against the real packages" (oracle/README.md)."""
import sys
import os

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_shim"))
import libconf      # noqa: E402  (the stand-in)
import magicattr    # noqa: E402

MANUAL_EXAMPLE = r'''
# Example application configuration file

version = "1.0";

application:
{
  window:
  {
    title = "My Application";
    size = { w = 640; h = 480; };
    pos = { x = 350; y = 250; };
  };

  list = ( ( "abc", 123, true ), 1.234, ( /* an empty list */ ) );

  books = ( { title  = "Treasure Island";
              author = "Robert Louis Stevenson";
              price  = 29.95;
              qty    = 5; },
            { title  = "Snow Crash";
              author = "Neal Stephenson";
              price  = 9.99;
              qty    = 8; } );

  misc:
  {
    pi = 3.141592654;
    bigint = 9223372036854775807L;
    columns = [ "Last Name", "First Name", "MI" ];
    bitmask = 0x1FC3;	// hex
  };
};
'''


def test_libconfig_manual_example():
    cfg = libconf.loads(MANUAL_EXAMPLE)
    assert cfg.version == "1.0" and cfg["version"] == "1.0"
    win = cfg.application.window
    assert win.title == "My Application" and (win.size.w, win.size.h) == (640, 480) and (win.pos.x, win.pos.y) == (350, 250)
    lst = cfg.application.list
    assert isinstance(lst, tuple) and lst == (("abc", 123, True), 1.234, ())
    books = cfg.application.books
    assert isinstance(books, tuple) and len(books) == 2
    assert books[0].title == "Treasure Island" and books[0].price == 29.95 and books[1].qty == 8 and books[1]["author"] == "Neal Stephenson"
    misc = cfg.application.misc
    assert misc.pi == 3.141592654 and misc.bigint == 9223372036854775807 and isinstance(misc.bigint, int)
    assert isinstance(misc.columns, list) and misc.columns == ["Last Name", "First Name", "MI"]
    assert misc.bitmask == 0x1FC3
    with pytest.raises(AttributeError):
        cfg.application.nothing_here
    assert "window" in cfg.application and "nothing_here" not in cfg.application


def test_libconfig_grammar_details_the_reference_relies_on():
    text = '''
    a : 1; b = 2        # ':' and '=' both assign, the terminator is optional
    s = "adjacent " "strings "
        "concatenate";  // (manual: "adjacent strings are automatically concatenated")
    t = TRUE; f = FaLsE;           /* booleans are case-insensitive */
    neg = -17; hexl = 0xFFL; flt = 1e3; flt2 = -.5;
    grp = { inner = ( 1, "two", [ 3, 4 ] ); };
    rep = 1; rep = 2;
    esc = "tab\\there\\n\\x41";
    '''
    c = libconf.loads(text)
    assert (c.a, c.b) == (1, 2) and c.s == "adjacent strings concatenate" and c.t is True and c.f is False
    assert c.neg == -17 and c.hexl == 255 and c.flt == 1000.0 and isinstance(c.flt, float) and c.flt2 == -0.5
    assert c.grp.inner == (1, "two", [3, 4]) and isinstance(c.grp.inner[2], list)
    assert c.rep == 2                                         # a repeated setting name: the later one stands (dict semantics)
    assert c.esc == "tab\there\nA"
    for bad in ("a = ;", "a = 1 b", "grp = { x = 1;", 'a = "unterminated'):
        with pytest.raises(Exception):
            libconf.loads(bad)


def test_magicattr_get_documented_forms():
    class Person:
        def __init__(self, name, age, friends=None):
            self.name, self.age, self.friends = name, age, friends or []
    jill, jack = Person("Jill", 29), Person("Jack", 28)
    bob = Person("Bob", 31, [jack, jill])
    bob.settings = {"style": {"width": 200}}
    assert magicattr.get(bob, "age") == 31                                  # README: "Nothing new"
    assert magicattr.get(bob, "friends[0].name") == "Jack"                  # README: "Lists too"
    assert magicattr.get(bob, "friends[1].age") == 29
    assert magicattr.get(bob, 'settings["style"]["width"]') == 200          # README: dictionary look-ups
    with pytest.raises(AttributeError):
        magicattr.get(bob, "friends[0].shoe_size")
    with pytest.raises(Exception):
        magicattr.get(bob, "friends[0].name; import os")                    # README: only attribute / subscript paths are evaluated
    cfg = libconf.loads('top = { groups = ( { cores = [ 1, 2 ]; } ); };')
    assert magicattr.get(cfg, "top.groups[0].cores[1]") == 2                # how nhd/TriadCfgParser.py:126-213 walks a config
