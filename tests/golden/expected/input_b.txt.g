Prev '\\sp' table:
Table:
a 1 1
Prev 'a' table:
Table:
a 1 1
b 1 0
Prev 'b' table:
Table:
b 1 0
c 1 1
Prev 'c' table:
Table:
b 1 0
d 1 1
Prev 'd' table:
Table:
c 1 1
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
	n1 [label="a"];
	n0 -- n1;
	n2;
	n2 [label="a"];
	n0 -- n2;
}
/* Prev 'a' tree: */
subgraph clusterG3 {
	label="Prev: a";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n3;
	n3 [label=""];
	n4;
	n4 [label="b"];
	n3 -- n4;
	n5;
	n5 [label="a"];
	n3 -- n5;
}
/* Prev 'b' tree: */
subgraph clusterG6 {
	label="Prev: b";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n6;
	n6 [label=""];
	n7;
	n7 [label="b"];
	n6 -- n7;
	n8;
	n8 [label="c"];
	n6 -- n8;
}
/* Prev 'c' tree: */
subgraph clusterG9 {
	label="Prev: c";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n9;
	n9 [label=""];
	n10;
	n10 [label="b"];
	n9 -- n10;
	n11;
	n11 [label="d"];
	n9 -- n11;
}
/* Prev 'd' tree: */
subgraph clusterG12 {
	label="Prev: d";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n12;
	n12 [label=""];
	n13;
	n13 [label="c"];
	n12 -- n13;
	n14;
	n14 [label="c"];
	n12 -- n14;
}
}
