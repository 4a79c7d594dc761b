"""Stand-in for the third-party `magicattr` package (deploy/requirements.txt: magicattr>=0.1.4) that
nhd/TriadCfgParser.py:17 imports.  TEST INFRASTRUCTURE ONLY (oracle/README.md).

magicattr.get(obj, "a.b[0].c") parses the path with `ast` and walks it with getattr / subscription; names,
attributes and constant subscripts are the supported nodes (anything else raises), which is all the reference uses."""
import ast


def _walk(obj, node):
    if isinstance(node, ast.Name):
        return getattr(obj, node.id)
    if isinstance(node, ast.Attribute):
        return getattr(_walk(obj, node.value), node.attr)
    if isinstance(node, ast.Subscript):
        sl = node.slice
        if isinstance(sl, ast.Index):          # Python < 3.9
            sl = sl.value
        if not isinstance(sl, ast.Constant):
            raise NotImplementedError("only constant subscripts are supported")
        return _walk(obj, node.value)[sl.value]
    raise NotImplementedError("unsupported path element %r" % type(node).__name__)


def get(obj, attr, **kwargs):
    tree = ast.parse(attr)
    if len(tree.body) != 1 or not isinstance(tree.body[0], ast.Expr):
        raise ValueError("invalid expression: %r" % (attr,))
    return _walk(obj, tree.body[0].value)
