"""Static check (no linter in the image): every name a function of a module loads as a GLOBAL exists in the module's namespace or in
builtins.  Catches a bare name left behind when module-level state moves into an object (dan_amd/ops.py: OpsContext).
usage: python tools/check_globals.py dan_amd.ops [dan_amd.trainer ...]"""
import ast
import builtins
import importlib
import sys
import symtable

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))


def walk(table, out):
    if table.get_type() == "function":
        for sym in table.get_symbols():
            if sym.is_global() and sym.is_referenced():
                out.add(sym.get_name())
    for child in table.get_children():
        walk(child, out)


bad = 0
for modname in sys.argv[1:]:
    mod = importlib.import_module(modname)
    src = open(mod.__file__).read()
    names = set()
    walk(symtable.symtable(src, mod.__file__, "exec"), names)
    missing = sorted(n for n in names if n not in mod.__dict__ and not hasattr(builtins, n))
    print("%s: %d global names referenced from functions, missing: %s" % (modname, len(names), missing or "none"))
    bad += len(missing)
sys.exit(1 if bad else 0)
