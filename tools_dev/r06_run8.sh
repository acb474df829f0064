#!/bin/bash
cd "$(dirname "$0")/.."
echo new; python tools_dev/dbg_fullgrid_fwd.py 2 2>&1 | tail -21 | cut -c1-260
echo old; GGA_SP_OFFSET_SUMS=0 python tools_dev/dbg_fullgrid_fwd.py 2 2>&1 | tail -21 | cut -c1-260
