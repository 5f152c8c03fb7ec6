#!/bin/bash
# no kernel of the shipped library may spill registers: see tools/spill_check.py
exec python "$(dirname "$0")/spill_check.py" "$@"
