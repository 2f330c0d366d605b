import os, subprocess, sys
for d in sys.argv[1:]:
    env = dict(os.environ, AFD_W2_DBG=d)
    out = subprocess.run([sys.executable, "tools_conv_bench.py", "conv3"], env=env, capture_output=True, text=True).stdout
    print("dbg", d, out.strip()[-40:])
