"""poor man's pyflakes (not installed here): names that are read in a function but bound nowhere (function, enclosing
functions, module, builtins).  usage: python scripts/undefined_names.py file.py ..."""
import builtins
import symtable
import sys


def walk(table, outer, path, out):
    bound = set(outer)
    for s in table.get_symbols():
        if s.is_assigned() or s.is_parameter() or s.is_imported() or s.is_namespace():
            bound.add(s.get_name())
    for s in table.get_symbols():
        n = s.get_name()
        if s.is_referenced() and n not in bound and not hasattr(builtins, n) and (s.is_global() or s.is_free() or not s.is_local()):
            out.append((path, n))
    for ch in table.get_children():
        walk(ch, bound, path + "." + ch.get_name(), out)


for f in sys.argv[1:]:
    src = open(f).read()
    top = symtable.symtable(src, f, "exec")
    out = []
    walk(top, set(), f, out)
    for p, n in out:
        print(p, "->", n)
