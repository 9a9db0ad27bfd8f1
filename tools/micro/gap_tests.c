#include <stdio.h>
#include <stdlib.h>
static int flt(int td,int qd,int mg){int gl=abs(td-qd);float gs=td>0?(float)qd/(float)td:1.0f;return gl<mg&&gs<5.0f&&gs>0.75f;}
static int itg(int td,int qd,int mg){int gl=abs(td-qd);return gl<mg&&qd<5*td&&qd>((3*td)>>2);}
int main(){long bad=0;
 for(int td=1;td<=6000;++td)for(int qd=0;qd<=5*td+3;++qd)for(int mi=0;mi<3;++mi){int mg=mi==0?2000:mi==1?2000000:2000000000;if(flt(td,qd,mg)!=itg(td,qd,mg))++bad;}
 for(long td=1;td<(1<<21);++td){int c[]={5*td-2,5*td-1,5*td,5*td+1,(3*td)/4-1,(3*td)/4,(3*td)/4+1,(3*td)/4+2,(3*td+3)/4,0,1,td,2147483647};
   for(int k=0;k<13;++k){int qd=c[k];if(qd<0)continue;if(flt(td,qd,2000000000)!=itg(td,qd,2000000000)){++bad;if(bad<5)printf("td %ld qd %d\n",td,qd);}}}
 printf("bad %ld\n",bad);return 0;}
