Prev '\\sp' table:
Table:
Z 1 1
graph G {
	packmode="cluster";
/* Prev '\\sp' tree: */
subgraph clusterG0 {
	label="Prev: \\sp";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n0;
	n0 [label=""];
	n1;
	n1 [label="Z"];
	n0 -- n1;
	n2;
	n2 [label="Z"];
	n0 -- n2;
}
}
