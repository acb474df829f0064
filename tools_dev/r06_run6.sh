#!/bin/bash
cd "$(dirname "$0")/.."
python tools_dev/dbg_fullgrid_grads.py bn 2 2>&1 | tail -1 | cut -c1-1500
python tools_dev/dbg_fullgrid_grads.py nobn 2 2>&1 | tail -1 | cut -c1-1500
GGA_SP_OFFSET_SUMS=0 python tools_dev/dbg_fullgrid_grads.py bn 2 2>&1 | tail -1 | cut -c1-1500
GGA_SP_OFFSET_SUMS=0 python tools_dev/dbg_fullgrid_grads.py nobn 2 2>&1 | tail -1 | cut -c1-1500
