Prev '\\sp' table:
Table:
Z 1 1
Prev 'Z' table:
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
/* Prev 'Z' tree: */
subgraph clusterG3 {
	label="Prev: Z";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n3;
	n3 [label=""];
	n4;
	n4 [label="Z"];
	n3 -- n4;
	n5;
	n5 [label="Z"];
	n3 -- n5;
}
}
