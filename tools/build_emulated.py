#!/usr/bin/env python3
"""builds the emulated library of tests/test_emulated_kernels.py (device code compiled for the host, one thread per lane) into
/tmp/emu_build and prints its path: for running tests/emu/*.py by hand (AFX_LIB=<path> python tests/emu/emulated_mfcc_sizes.py 10)"""
import os, pathlib, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_emulated_kernels as t


class Factory:
    def mktemp(self, name):
        d = "/tmp/emu_build"
        os.makedirs(d, exist_ok=True)
        return pathlib.Path(d)


print(t.emulated._get_wrapped_function()(Factory()))
