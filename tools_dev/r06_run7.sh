#!/bin/bash
cd "$(dirname "$0")/.."
echo new; python tools_dev/dbg_fullgrid_ops.py 2 2>&1 | tail -22 | cut -c1-260
echo old; GGA_SP_OFFSET_SUMS=0 python tools_dev/dbg_fullgrid_ops.py 2 2>&1 | tail -22 | cut -c1-260
