"""Every name a function of a module reads as a global must exist in the module (or be a builtin): a static pass over the modules of the
replay engine (no linter in the image).  `python tools/check_globals.py` -> exit code 1 and the missing names if any."""
import builtins
import importlib
import os
import symtable
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
MODULES = sys.argv[1:] or ["lavis.compression.pruners.calibration", "lavis.compression.pruners.replay_state", "lavis.compression.pruners.replay_padding",
                           "lavis.compression.pruners.replay_towers", "lavis.compression.pruners.replay_capture", "vlmc.forward", "vlmc.ops"]
bad = 0
for name in MODULES:
    mod = importlib.import_module(name)
    src = open(mod.__file__).read()
    top = symtable.symtable(src, mod.__file__, "exec")
    have = set(dir(mod)) | set(dir(builtins))

    def walk(t, path):
        global bad
        for s in t.get_symbols():
            if t.get_type() != "module" and s.is_global() and s.is_referenced() and s.get_name() not in have:
                print(f"{name}: {'.'.join(path)} reads global {s.get_name()!r} that the module does not define")
                bad += 1
        for c in t.get_children():
            walk(c, path + [c.get_name()])
    walk(top, [])
print("checked", len(MODULES), "modules:", "ok" if not bad else f"{bad} missing")
sys.exit(1 if bad else 0)
