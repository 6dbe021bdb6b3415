#include "plan.hpp"
int main(){ for (uint32_t p : {31u,127u,933u,9941u,100003u,9815459u,57885161u,136279841u,205271257u,332000003u,600000001u}) { auto pl = mi355::make_plan(p, nullptr, p < 1000000); printf("p=%u %s L1=%u lds_front=%zu lds_mid=%zu tiles=%zu\n", p, pl.describe().c_str(), pl.L1, pl.lds_front, pl.lds_mid, pl.tiles()); } }
