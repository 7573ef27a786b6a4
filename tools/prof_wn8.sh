#!/bin/bash
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc EXTRA="-DFACPPG_WN8_PROF" 2>/dev/null >/dev/null
python tools/prof_wn8.py ${1:-200} 2>&1 | tail -9
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc 2>/dev/null >/dev/null
