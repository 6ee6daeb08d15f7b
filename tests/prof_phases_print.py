# prints the differences between the phase stamps tests/prof_phases.py wrote to gpurun_out/phases_now.txt
for l in open("gpurun_out/phases_now.txt"):
    if "k_dense stamps" in l:
        st = eval(l.split("stamps(ticks)")[1].split("wall_ns")[0])
        print("dense 3->4", st[4] - st[3], "4->5 chol", st[5] - st[4], "5->6 backsub", st[6] - st[5], "6->7 out", st[7] - st[6], "4->8 load", st[8] - st[4],
              "| panel0: 8->10", st[10] - st[8], "10->11", st[11] - st[10], "11->12", st[12] - st[11])
    if "k_linearize stamps" in l:
        st = eval(l.split("stamps(ticks)")[1].split("wall_ns")[0])
        print("linearize 0->1 prologue", st[1] - st[0], "1->9 roles", st[9] - st[1], l.split("wall_ns")[1].strip())
