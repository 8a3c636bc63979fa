#!/bin/bash
# A/B two builds of libmeant_hip.so on the same box: $1 = path of the alternative .so (copied next to the current one)
python tools/probe_attn.py 2>&1 | grep "G=" | sed 's/^/NEW /'
cp meant_amd/libmeant_hip.so /tmp/new.so && cp "$1" meant_amd/libmeant_hip.so
python tools/probe_attn.py 2>&1 | grep "G=" | sed 's/^/OLD /'
cp /tmp/new.so meant_amd/libmeant_hip.so
python tools/probe_attn.py 2>&1 | grep "G=" | sed 's/^/NEW /'
