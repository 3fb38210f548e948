for i in 1 2 3; do for d in _ab_eb5a485 .; do echo "== $d $(cd $d && python profiles/conv_patch_probe.py 6 8 2>&1 | grep conv | awk '{print $(NF-3)}' | tr '\n' ' ')"; done; done
