# Does a second process on the GPU matter?  (a) alone, (b) an idle neighbour holding a context, (c) a neighbour doing small kernels, (d) the reference server
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== (a) alone"; FLOW_NO_REF=1 timeout 300 python tests/triage/fuzz_flow.py 175 186 6 2>&1 | grep -c BAD
echo "== (b) idle neighbour"
python -c "import torch,time; x=torch.zeros(1,device='cuda'); torch.cuda.synchronize(); print('neighbour up',flush=True); time.sleep(120)" & NB=$!
sleep 15
FLOW_NO_REF=1 timeout 300 python tests/triage/fuzz_flow.py 175 186 6 2>&1 | grep -c BAD
kill $NB; wait $NB 2>/dev/null
echo "== (c) busy neighbour"
python -c "
import torch,time
x=torch.zeros(1<<20,device='cuda'); t=time.time()
while time.time()-t<120:
    for _ in range(50): x.add_(1)
    torch.cuda.synchronize(); time.sleep(0.01)
" & NB=$!
sleep 15
FLOW_NO_REF=1 timeout 300 python tests/triage/fuzz_flow.py 175 186 6 2>&1 | grep -c BAD
kill $NB; wait $NB 2>/dev/null
echo "== (d) reference server"; timeout 300 python tests/triage/fuzz_flow.py 175 186 6 2>&1 | grep -c BAD
