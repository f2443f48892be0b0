#!/bin/bash
# Several "key=value[,key=value]" option sets through tools/ab_opts.sh on the in-tree library:  tools/sweep_many.sh "pw_refill=8" "pw_leaf_go=24,pw_refill=16" ... [-- bench args]
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("default|$1"); shift; done
tools/ab_opts.sh "default|" "${sets[@]}" "$@"
