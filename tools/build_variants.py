"""Builds kernel variants of libmeao_hip.so next to the product library for A/B runs on one GPU box:
    python tools/build_variants.py name=-DFLAG[,-DFLAG2] ...      ->  miniengineao_amd/lib/variants/libmeao_<name>.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from miniengineao_amd import build
out_dir = os.path.join(build.LIB_DIR, "variants")
os.makedirs(out_dir, exist_ok=True)
def one(spec):
    name, _, flags = spec.partition("=")
    path = os.path.join(out_dir, f"libmeao_{name}.so")
    build.build_lib(force=True, extra_flags=[f for f in flags.split(",") if f], out_path=path)
    return path
with ThreadPoolExecutor(4) as ex:
    for p in ex.map(one, sys.argv[1:]):
        print(p)
