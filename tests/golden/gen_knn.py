"""Golden vectors for the EdgeConv k-NN graph: the reference's OWN knn() / get_graph_feature()
(models/sparenet_generator.py:852-906), executed here on the CPU (its torch fallback branch: there is
no CUDA and no KNN_CUDA wheel in this container).  The two functions' text is read from
/root/reference at run time and exec'd; nothing of it is stored in this repository."""
import os

import numpy as np
import torch

REF = "/root/reference/models/sparenet_generator.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_functions():
    lines = open(REF).read().split("\n")
    text = "\n".join(lines[851:906])           # def knn ... end of get_graph_feature
    ns = {"torch": torch}
    exec(compile(text, REF, "exec"), ns)
    return ns["knn"], ns["get_graph_feature"]


def main():
    knn, get_graph_feature = load_reference_functions()
    for name, b, c, n, k, seed in [("knn_2x16x200_k8", 2, 16, 200, 8, 0), ("knn_1x3x500_k20", 1, 3, 500, 20, 1)]:
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(b, c, n, generator=g)
        idx = knn(x, k)
        feat = get_graph_feature(x, k=k, idx=idx.clone())
        np.savez_compressed(os.path.join(HERE, name + ".npz"), x=x.numpy(), k=np.int32(k),
                            idx=idx.numpy(), feature=feat.numpy(),
                            provenance=np.array("reference knn()/get_graph_feature() CPU branch, imported text"))
        print(name, idx.shape, feat.shape)


if __name__ == "__main__":
    main()
