#!/usr/bin/env python3
"""INTEGRATION.md's tuning table, from the library's own (mtg_tuning_describe): python scripts/print_tuning_table.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mindthegap_amd as m
kinds = {"cap": "capacities and sizes", "ab": "A/B hooks: a measured alternative kept runnable", "test": "test hooks: force a rare path", "diag": "diagnostics"}
for kind, title in kinds.items():
    print("| **%s** | | |" % title)
    for e in m.tuning():
        if e["kind"] == kind:
            print("| `%s` | %s | %s |" % (e["name"], ("`%s`" % e["default"]) if e["default"] else "—", e["what"].replace("|", "\\|")))
