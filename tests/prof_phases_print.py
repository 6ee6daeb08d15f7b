for l in open("gpurun_out/phases_now.txt"):
    if "k_dense stamps" in l:
        st=eval(l.split("stamps(ticks)")[1].split("wall_ns")[0])
        print("dense 3->4", st[4]-st[3], "4->5 chol", st[5]-st[4], "5->6 backsub", st[6]-st[5], "6->7 out", st[7]-st[6], "4->8 load", st[8]-st[4], "| panel0: 8->10", st[10]-st[8], "10->11", st[11]-st[10], "11->12", st[12]-st[11], "| panel@80: 13->15", st[15]-st[13], "15->16", st[16]-st[15], "16->17", st[17]-st[16])
